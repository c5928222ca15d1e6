#!/usr/bin/env python3
"""Exhausted-Krylov-space probe: the n = 2000 alternating ring of examples/drop_in.cpp, two lowest
pairs, offset -3, fixed splitmix start vector.  Runs the host-callback path and the device-CSR path with LL_ITER_TRACE
and prints values / counts; the traces land in gpurun_out/ring_<path>_<fuse>.trace."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out_dir = os.path.join(ROOT, "gpurun_out")
os.makedirs(out_dir, exist_ok=True)

import lambda_lanczos_amd as L  # noqa: E402
from lambda_lanczos_amd import generators as G  # noqa: E402

n = 2000
rows, cols, vals = [], [], []
for i in range(n):
    rows += [i, i, i]
    cols += [i, (i + 1) % n, (i + n - 1) % n]
    vals += [0.3 if i % 2 else -0.3, -1.0, -1.0]
csr = G.coo_to_csr(n, rows, cols, np.array(vals))
import scipy.sparse as sp  # noqa: E402

A = sp.csr_matrix((csr[2], csr[1], csr[0]), shape=(n, n))


def mv(a, b):
    b += A @ a


seeds = [int(s) for s in os.environ.get("SEEDS", "1").split(",")]
for fuse in os.environ.get("FUSE", "2,0").split(","):
    for path in ("cb", "csr"):
        for seed in seeds:
            tr = os.path.join(out_dir, "ring_%s_f%s_s%d.trace" % (path, fuse, seed))
            if os.path.exists(tr):
                os.remove(tr)
            os.environ["LL_ITER_TRACE"] = tr
            os.environ["LL_FUSE_LAUNCHES"] = fuse
            ctx = L.Context(0)
            init = G.start_vector(n, seed)
            op = L.HostOperator(ctx, mv, n) if path == "cb" else L.CsrOperator(ctx, *csr)
            eng = L.LambdaLanczos(op, n, False, 2)
            eng.eigenvalue_offset = -3.0
            eng.init_vector = lambda v, *_: v.__setitem__(slice(None), init)
            v, x = eng.run()
            st = eng.last_stats
            print("fuse=%s %s seed=%d E0=%.12f E1=%.12f counts=%s second_passes=%s lagged=%s calls=%s" % (
                fuse, path, seed, v[0], v[1], eng.getIterationCounts(), st.get("second_passes"), st.get("lagged_iterations"),
                getattr(op, "calls", None)), flush=True)
            op.close()
            ctx.close()
