#!/usr/bin/env python3
"""Time-to-convergence on the GPU with the reference's DEFAULT settings (nroot = 5, eps = 1e3*eps_machine) on the
BASELINE configurations, with eigenpair checks (reported number (3)).

    python tests/convergence_run.py c3 [tridiag_mode]      # random symmetric n=1e7, largest eigenpair
    python tests/convergence_run.py c2 [tridiag_mode]      # 5-point Laplacian n=1e6, smallest, offset -8

Lives under tests/ because its optional second leg (DEMO_ORACLE_THREADS / DEMO_USE_REFERENCE) runs the same problem through
the CPU checkers (oracle/, oracle/_ref) — test infrastructure that only tests/, smoke() and bench.py's cpu_baseline may use.
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lambda_lanczos_amd as L  # noqa: E402
from lambda_lanczos_amd import generators as G  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
mode = int(sys.argv[2]) if len(sys.argv) > 2 else None   # None: the library default (LL_TRIDIAG_AUTO)
ctx = L.Context(0)
for kv in filter(None, os.environ.get("DEMO_TUNING", "").split(",")):   # per-context settings for A/B runs: key=value,key=value
    ctx.set_tuning(*kv.split("=", 1))
if wl == "c3":
    n = int(os.environ.get("DEMO_N", "10000000"))
    csr = G.randsym(n)
    find_max, offset = True, 0.0
else:
    side = int(os.environ.get("DEMO_N", "1000"))
    n = side * side
    csr = G.laplace2d(side)
    find_max, offset = False, -8.0
op = L.CsrOperator(ctx, *csr)
init = G.start_vector_fast(n, 1)
eng = L.LambdaLanczos(op, n, find_max, 1)
eng.eigenvalue_offset = offset
if mode is not None:
    eng.tridiag_mode = mode
eng.init_vector = lambda v, *_: np.copyto(v, init)
ctx.set_profiling(True)
t0 = time.time()
vals, vecs = eng.run()
wall = time.time() - t0
first_stats = dict(eng.last_stats)
# the same call again: the basis and the work vectors of the first call are kept by the context's cache, so this one shows what
# the first one spent on device allocations (boxes of the pool differ by 0.2 s there)
t0 = time.time()
vals2, vecs2 = eng.run()
wall2 = time.time() - t0
assert np.array_equal(vals, vals2) and np.array_equal(vecs[0], vecs2[0])
v = vecs[0]
xd, yd = ctx.to_device(v), ctx.empty(n)
L.spmv(op, xd, yd)
res = float(np.linalg.norm(yd.get() - vals[0] * v))
out = {"workload": wl, "n": n, "tridiag_mode": int(eng.tridiag_mode), "iterations": eng.getIterationCounts(), "eigenvalue": float(vals[0]),
       "residual_norm": res, "wall_s": wall, "wall_s_second_call": wall2, "second_call_bit_identical": True,
       "stats": first_stats, "stats_second_call": eng.last_stats}
if wl == "c2":
    out["analytic_lambda_min"] = G.laplace2d_lambda_min(int(round(n ** 0.5)))
print(json.dumps(out), flush=True)
if os.environ.get("DEMO_ORACLE_THREADS"):
    # the same run through the CPU oracle (pinned to the real reference; OpenMP variant to keep it to minutes)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib

    if os.environ.get("DEMO_USE_REFERENCE"):      # the REAL reference headers (single thread, oracle/_ref/libref.so)
        orc, th = oracle_lib.reference(), 1
    else:
        orc = oracle_lib.oracle()
        th = orc.set_threads(int(os.environ["DEMO_ORACLE_THREADS"]))
    t0 = time.time()
    r = orc.lanczos(csr, init, find_max, offset=offset, trace=False)
    cpu_wall = time.time() - t0
    ov = abs(np.vdot(r["eigenvectors"][0], v))
    print(json.dumps({"checker": "reference" if os.environ.get("DEMO_USE_REFERENCE") else "oracle", "oracle_threads": th, "oracle_iterations": r["iter_counts"], "oracle_eigenvalue": float(r["eigenvalues"][0]),
                      "oracle_wall_s": cpu_wall, "d_lambda": float(abs(r["eigenvalues"][0] - vals[0])),
                      "one_minus_overlap": float(1 - ov), "gpu_wall_s": wall, "speedup": cpu_wall / wall}), flush=True)
