#!/usr/bin/env python3
"""A/B the SpMV kernels and the tiled-COO geometry in ONE process (interleaved rounds, median and min reported —
cdna_hip_programming.md rule 24).  Usage: python tools/spmv_sweep.py [--workload c3|c3band|c2] [--n N]"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lambda_lanczos_amd as L  # noqa: E402
from lambda_lanczos_amd import generators as G  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="c3")
ap.add_argument("--n", type=int, default=0)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--nocheck", action="store_true")
ap.add_argument("--geoms", default="0:0,13021:13021,13021:9766,9766:19456,13021:6511")
a = ap.parse_args()

if a.workload in ("c3", "c3band"):
    n = a.n or 10_000_000
    csr = G.randsym(n, band=65536 if a.workload == "c3band" else 0)
elif a.workload == "c2":
    side = a.n or 1000
    n = side * side
    csr = G.laplace2d(side)
nnz = int(csr[0][-1])
bytes_ = 12 * nnz + 4 * (n + 1) + 16 * n
ctx = L.Context(0)
x = G.start_vector_fast(n, 1)
xd, yd = ctx.to_device(x / np.linalg.norm(x)), ctx.empty(n)
variants = {}
os.environ["LL_SPMV_KERNEL"] = "csr"
variants["csr_stream"] = L.CsrOperator(ctx, *csr)
os.environ.pop("LL_SPMV_KERNEL")
for g in a.geoms.split(","):
    os.environ["LL_SPMV_KERNEL"] = "pb"
    cb, rb = (g.split(":") + ["0"])[:2]
    for key, val in (("LL_PB_COL_BLOCK", cb), ("LL_PB_ROW_BLOCK", rb)):
        if int(val):
            os.environ[key] = val
        else:
            os.environ.pop(key, None)
    variants["pb_col%s_row%s" % (cb, rb)] = L.CsrOperator(ctx, *csr)
ref = None
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
        if rnd == 0:
            y = yd.get()
            if ref is None:
                ref = y
            else:
                assert a.nocheck or np.max(np.abs(y - ref)) <= 1e-12 * np.max(np.abs(ref)), name
for name, t in times.items():
    t = sorted(t)
    med, mn = t[len(t) // 2], t[0]
    print(json.dumps({"variant": name, "ms_median": med, "ms_min": mn, "GBps_median": bytes_ / med / 1e6,
                      "frac_of_8TBps": bytes_ / med / 1e6 / 8000.0}))
