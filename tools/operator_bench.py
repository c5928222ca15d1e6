#!/usr/bin/env python3
"""Apply rate of the non-CSR operators: dense row block, matrix-free lattice.  HIP events on the library
stream, median of 5 rounds of 10 applies.   python tools/operator_bench.py"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lambda_lanczos_amd as L  # noqa: E402

ctx = L.Context(0)


def rate(op, n, dtype, bytes_):
    x = np.ones(n, dtype=dtype)
    xd, yd = ctx.to_device(x), ctx.empty(n, dtype)
    ms = []
    for _ in range(6):
        L.spmv(op, xd, yd)
        ctx.synchronize()
        ctx.timer_start()
        for _ in range(10):
            L.spmv(op, xd, yd, want_dot=False)
        ms.append(ctx.timer_stop() / 10)
    m = sorted(ms[1:])[2]
    return {"ms": m, "GBps": bytes_ / m / 1e6, "frac_of_8TBps": bytes_ / m / 1e6 / 8000.0}


out = {}
for n in (4096, 16384):
    rng = np.random.default_rng(1)
    a = rng.standard_normal((n, n))
    op = L.DenseOperator(ctx, a)
    out["dense_f64_n%d" % n] = dict(rate(op, n, np.float64, 8 * n * n + 16 * n), algorithmic_bytes="8 n^2 + 16 n")
    op.close()
    del a
for dims in ([4096, 4096], [256, 256, 256], [16777216]):
    n = int(np.prod(dims))
    op = L.StencilOperator(ctx, dims, diag=2.0 * len(dims), hop=-1.0, periodic=True)
    out["lattice_f64_" + "x".join(map(str, dims))] = dict(rate(op, n, np.float64, 16 * n), algorithmic_bytes="16 n")
    op.close()
n = 4096 * 4096
op = L.StencilOperator(ctx, [4096, 4096], diag=0.0, hop=[-1.0 + 0.5j, -1.0], periodic=True, dtype=np.complex128,
                       onsite=np.cos(np.arange(n)))
out["lattice_c128_4096x4096_onsite"] = dict(rate(op, n, np.complex128, 40 * n), algorithmic_bytes="32 n + 8 n")
op.close()
print(json.dumps(out, indent=1))
