"""Why is the first CSR operator of a process 3-6 % slower than later ones ("position noise")?  Creates the
config-3 operator several times in one process (closing each before the next) and times its SpMV; then does the same
with a 4 GiB device allocation made and freed before every creation.
  python tools/position_probe.py > gpurun_out/position_probe.txt"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import lambda_lanczos_amd as L  # noqa: E402
from lambda_lanczos_amd import generators as G  # noqa: E402

os.environ["LL_SPMV_KERNEL"] = "pb"
ctx = L.Context(0)
n = 10_000_000
csr = G.randsym(n)
x = ctx.to_device(G.start_vector_fast(n, 1) / np.sqrt(n / 3.0))
y = ctx.empty(n)


def timed(op):
    best = []
    for _ in range(3):
        L.spmv(op, x, y)
        ctx.synchronize()
        ctx.timer_start()
        for _ in range(20):
            L.spmv(op, x, y)
        best.append(ctx.timer_stop() / 20)
    return sorted(best)[1]


for label, pre in (("plain", False), ("4 GiB allocated and freed before each creation", True), ("plain again", False)):
    out = []
    for i in range(4):
        if pre:
            d = L.DeviceArray(ctx, (1 << 29,), np.float64)
            d.free()
        op = L.CsrOperator(ctx, *csr)
        out.append(round(timed(op), 4))
        op.close()
    print(label, out, flush=True)
# keep two operators alive at once: does the second one differ?
a = L.CsrOperator(ctx, *csr)
b = L.CsrOperator(ctx, *csr)
print("two alive: first", round(timed(a), 4), "second", round(timed(b), 4), "first again", round(timed(a), 4))
