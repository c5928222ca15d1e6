"""Placement draws of config 3's PB image (capi.cpp tune_pb_placement) in one process: LL_PB_PLACEMENTS draws, each timed with the
real kernels; LL_PB_PLACEMENT_TRACE=1 prints every draw.  Run several processes on one box to separate what varies per
allocation from what varies per process."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("LL_PB_PLACEMENTS", "10")
os.environ["LL_PB_PLACEMENT_TRACE"] = "1"
import numpy as np  # noqa: E402
import lambda_lanczos_amd as L  # noqa: E402
from lambda_lanczos_amd import generators as G  # noqa: E402

n = 10_000_000
ctx = L.Context(0)
csr = G.randsym(n)
op = L.CsrOperator(ctx, *csr)
x = G.start_vector_fast(n, 1)
xd, yd = ctx.to_device(x / np.linalg.norm(x)), ctx.empty(n)
ts = []
for _ in range(3):
    L.spmv(op, xd, yd)
    ctx.synchronize()
    ctx.timer_start()
    for _ in range(20):
        L.spmv(op, xd, yd)
    ts.append(ctx.timer_stop() / 20)
print("kept image, 3 rounds of 20 launches: %s ms" % ["%.4f" % t for t in ts], flush=True)
