"""A/B of the tiled SpMV kernel's two locality switches on the banded config 3, same process, interleaved rounds of 20 launches
by HIP events: the per-context settings tl_xcd (one contiguous eighth of the row blocks per XCD against launch order; read per launch) and tl_walk
(a row block's tiles by column index modulo the longest tile list against ascending order; read at operator creation).
    python tools/tl_xcd_probe.py
    LL_TL_PROBE_ONLY=<xcd><walk> (e.g. 11, 10, 01, 00) under rocprofv3 --pmc FETCH_SIZE: one combination, for its traffic"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lambda_lanczos_amd as L  # noqa: E402
from lambda_lanczos_amd import generators as G  # noqa: E402

n = 10_000_000
only = os.environ.get("LL_TL_PROBE_ONLY")
ctx = L.Context(0)
csr = G.randsym(n, band=65536)
x = G.start_vector_fast(n, 1)
xd = ctx.to_device(x / np.linalg.norm(x))
ops, ys = {}, {}
for walk in ("1", "0"):
    if only is not None and only[1] != walk:
        continue
    ctx.set_tuning("tl_walk", walk)
    ops[walk] = L.CsrOperator(ctx, *csr, kernel=L.capi.SPMV_TILED)
    ys[walk] = ctx.empty(n)
combos = [only] if only is not None else ["11", "01", "10", "00"]
res = {c: [] for c in combos}
for rnd in range(int(os.environ.get("LL_TL_PROBE_ROUNDS", "1" if only is not None else "5"))):
    for c in combos:
        ctx.set_tuning("tl_xcd", c[0])
        op, yd = ops[c[1]], ys[c[1]]
        L.spmv(op, xd, yd)
        ctx.synchronize()
        ctx.timer_start()
        for _ in range(20):
            L.spmv(op, xd, yd)
        res[c].append(ctx.timer_stop() / 20)
same = None
if len(ys) == 2:
    same = bool(np.array_equal(ys["1"].get(), ys["0"].get()))
print(json.dumps({"ms_by_<xcd><walk>": res, "walk_orders_bit_identical": same}))
