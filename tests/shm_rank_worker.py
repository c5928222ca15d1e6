"""One rank of tests/test_gpu_multirank.py: several of these processes share the single GPU of the test box and talk
through the library's host-staged test communicator (LL_COMM_BACKEND=shm) — the sharded engine with real HIP kernels
and N > 1 ranks.  argv: rank world shm_name out_dir"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import lambda_lanczos_amd as L  # noqa: E402
from lambda_lanczos_amd import generators as G  # noqa: E402
from util import c2list  # noqa: E402


def main():
    rank, world, name, out_dir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    ctx = L.Context(0)
    ctx.init_comm(name.encode() + b"\0" * (128 - len(name)), rank, world)
    res = {}
    # --- real symmetric, two roots (restart pass with a locked, sharded eigenvector), both SpMV kernels
    n = 9001
    rb, nl = ctx.partition(n)
    csr = G.randsym(n, row_begin=rb, n_local=nl)
    init = G.start_vector(nl, 1, np.float64, rb)
    for kind, label in ((L.capi.SPMV_CSR_STREAM, "csr"), (L.capi.SPMV_PB, "pb")):
        op = L.CsrOperator(ctx, *csr, n_cols=n, row_begin=rb)
        op.select_spmv(kind)
        eng = L.LambdaLanczos(op, n, True, 2)
        eng.max_iteration = 120        # bounded: the host-staged test transport is slow
        eng.init_vector = lambda v, row_begin: np.copyto(v, G.start_vector(v.shape[0], 1, np.float64, row_begin))
        vals, vecs = eng.run()
        res["randsym_" + label] = {"row_begin": rb, "n_local": nl, "vals": vals.tolist(), "vecs": [v.tolist() for v in vecs],
                                   "iters": eng.getIterationCounts(), "alpha": eng.last_alpha.tolist()}
        # SpMV alone on a known vector
        xd, yd = ctx.to_device(init), ctx.empty(nl)
        dot = L.spmv(op, xd, yd, offset=0.5, want_dot=True)
        res["spmv_" + label] = {"y": yd.get().tolist(), "dot": dot}
        op.close()
    # --- Laplacian, smallest, offset
    side = 24
    n2 = side * side
    rb2, nl2 = ctx.partition(n2)
    lap = L.CsrOperator(ctx, *G.laplace2d(side, rb2, nl2), n_cols=n2, row_begin=rb2)
    e2 = L.LambdaLanczos(lap, n2, False, 1)
    e2.eigenvalue_offset = -8.0
    e2.init_vector = lambda v, row_begin: np.copyto(v, G.start_vector(v.shape[0], 1, np.float64, row_begin))
    v2, x2 = e2.run()
    res["laplace"] = {"vals": v2.tolist(), "vecs": [x2[0].tolist()], "iters": e2.getIterationCounts(), "row_begin": rb2}
    lap.close()
    # --- complex Hermitian torus: exp(-iH)v, sharded input/output
    N = 24
    n3 = N * N
    rb3, nl3 = ctx.partition(n3)
    top = L.CsrOperator(ctx, *G.torus(N, rb3, nl3), n_cols=n3, row_begin=rb3)
    inp = G.start_vector(nl3, 1, np.complex128, rb3)
    out, it = L.Exponentiator(top, n3).run(-1j, inp)
    res["torus_expo"] = {"out": c2list(out), "itern": it, "row_begin": rb3}
    top.close()
    with open(os.path.join(out_dir, "rank%d.json" % rank), "w") as f:
        json.dump(res, f)
    ctx.close()


if __name__ == "__main__":
    main()
