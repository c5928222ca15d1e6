"""Does the placement of the Krylov-basis slabs matter like the placement of the PB image does?  One process,
several fresh contexts one after the other (each allocates its own 4 GiB slabs), the same 100-iteration window of an n = 1e7
problem with a cheap operator; prints the device time of the Gram-Schmidt sweeps per context.

    python tools/slab_placement_probe.py [contexts] [n]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("OMP_WAIT_POLICY", "passive")
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
import numpy as np

import lambda_lanczos_amd as L
from lambda_lanczos_amd import generators as G


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
    init = G.start_vector_fast(n, 1)
    keep = []
    for r in range(reps):
        ctx = L.Context(0)
        ctx.set_profiling(True)
        op = L.StencilOperator(ctx, [n], diag=2.0, hop=-1.0)
        eng = L.LambdaLanczos(op, n, True, 1)
        eng.max_iteration = 100
        eng.eps = 0.0
        d_init = ctx.to_device(init)
        eng.init_vector = d_init
        times = []
        for _ in range(3):
            eng.run()
            times.append(eng.last_stats["seconds_orth"])
        print("context %d: Gram-Schmidt device time per window %.2f %.2f %.2f ms -> %.2f TB/s" % (
            r, times[0] * 1e3, times[1] * 1e3, times[2] * 1e3, 436.0e9 / min(times) / 1e12), flush=True)
        # keep every other context alive so that the next one's slabs land somewhere else
        if r % 2 == 0:
            keep.append((ctx, op, eng, d_init))
        else:
            op.close()
            ctx.close()


if __name__ == "__main__":
    main()
