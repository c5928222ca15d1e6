#!/usr/bin/env python3
"""A/B SpMV kernel variants in ONE process (interleaved rounds, median and min reported — cdna_hip_programming.md
rule 24).  Every variant is an operator created under its own environment (the kernel knobs are fixed per operator at
creation), e.g.

    python tools/spmv_sweep.py --workload c3 \
        --variants "csr:LL_SPMV_KERNEL=csr;pb:LL_SPMV_KERNEL=pb;pb_atomic:LL_SPMV_KERNEL=pb,LL_PB_PHASE2=atomic"

Also reports, per variant, whether repeated launches give bit-identical y (run-to-run determinism) and the largest
deviation from the first variant.  One JSON line per variant."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lambda_lanczos_amd as L  # noqa: E402
sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import sync_hooks  # noqa: E402
from lambda_lanczos_amd import generators as G  # noqa: E402

DEFAULT = ";".join([
    "csr_stream:LL_SPMV_KERNEL=csr",
    "pb_fixed_default:LL_SPMV_KERNEL=pb",
    "pb_ordered:LL_SPMV_KERNEL=pb,LL_PB_PHASE2=ordered",
    "pb_atomic:LL_SPMV_KERNEL=pb,LL_PB_PHASE2=atomic",
])

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="c3")
ap.add_argument("--n", type=int, default=0)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--variants", default=DEFAULT)
ap.add_argument("--repeat-check", type=int, default=4, help="launches compared bit for bit per variant")
a = ap.parse_args()

if a.workload in ("c3", "c3band"):
    n = a.n or 10_000_000
    csr = G.randsym(n, band=65536 if a.workload == "c3band" else 0)
elif a.workload == "c2":
    side = a.n or 1000
    n = side * side
    csr = G.laplace2d(side)
else:
    raise SystemExit("workload: c3 | c3band | c2")
nnz = int(csr[0][-1])
bytes_ = 12 * nnz + 4 * (n + 1) + 16 * n
ctx = L.Context(0)
x = G.start_vector_fast(n, 1)
xd, yd = ctx.to_device(x / np.linalg.norm(x)), ctx.empty(n)

variants, envs = {}, {}
for spec in a.variants.split(";"):
    name, _, kv = spec.partition(":")
    env = dict(item.split("=", 1) for item in kv.split(",") if item)
    saved = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        ctx.reload_env()   # the user-facing switches are read once per context ...
        sync_hooks(ctx)    # ... the geometry overrides / hooks are per-context settings (tests/util.py HOOK_KEYS)
        variants[name] = L.CsrOperator(ctx, *csr)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        ctx.reload_env()
        sync_hooks(ctx)
    envs[name] = env

ref = None
info = {}
for name, op in variants.items():
    ys = []
    for _ in range(max(1, a.repeat_check)):
        L.spmv(op, xd, yd)
        ys.append(yd.get())
    same = all(np.array_equal(ys[0], y) for y in ys[1:])
    if ref is None:
        ref = ys[0]
    info[name] = {"bit_identical_over_%d_launches" % len(ys): bool(same),
                  "max_abs_dev_from_first_variant": float(np.max(np.abs(ys[0] - ref))),
                  "max_abs_y": float(np.max(np.abs(ref)))}
times = {k: [] for k in variants}
for rnd in range(a.rounds + 1):
    for name, op in variants.items():
        L.spmv(op, xd, yd)
        ctx.synchronize()
        ctx.timer_start()
        for _ in range(a.reps):
            L.spmv(op, xd, yd)
        ms = ctx.timer_stop() / a.reps
        if rnd:
            times[name].append(ms)
for name, t in times.items():
    t = sorted(t)
    med, mn = t[len(t) // 2], t[0]
    out = {"variant": name, "env": envs[name], "n": n, "nnz": nnz, "ms_median": med, "ms_min": mn,
           "GBps_median": bytes_ / med / 1e6, "frac_of_8TBps": bytes_ / med / 1e6 / 8000.0}
    out.update(info[name])
    print(json.dumps(out), flush=True)
