"""One rank of tests/test_gpu_multirank.py: several of these processes share the single GPU of the test box and talk
through the host-staged test transport (LL_COMM_PLUGIN=tests/transport/_build/libll_shm_transport.so) — the sharded engine with real HIP kernels
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
from util import install_hook_sync  # noqa: E402
from lambda_lanczos_amd import generators as G  # noqa: E402
from util import c2list  # noqa: E402


install_hook_sync()   # the harness's hook settings (util.HOOK_KEYS in os.environ) -> every context of this process


def main():
    rank, world, name, out_dir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    os.environ["LL_SPMV_KEEP_BOTH"] = "1"   # this worker switches kernels on live operators (select_spmv)
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
                                   "iters": eng.getIterationCounts(), "alpha": eng.last_alpha.tolist(),
                                   "lagged": int(eng.last_stats["lagged_iterations"]),
                                   "pair": int(eng.last_stats["pair_iterations"])}
        if label == "csr":   # run_iteration with a sharded orthogonalizeTo list: the first eigenvector is locked
            eng.max_iteration = 60
            rv, rx, rit = eng.run_iteration(2, [vecs[0]])
            res["run_iteration"] = {"vals": rv.tolist(), "vecs": [v.tolist() for v in rx], "itern": rit}
            eng.max_iteration = 120
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
    # ... and once more through the 2-D tiled kernel (complex entries, sharded: two launches per product)
    ctx.set_tuning("tl_force", "1")
    ctx.set_tuning("pb_row_block", "40")
    top2 = L.CsrOperator(ctx, *G.torus(N, rb3, nl3), n_cols=n3, row_begin=rb3, kernel=L.capi.SPMV_TILED)
    out2, it2 = L.Exponentiator(top2, n3).run(-1j, inp)
    res["torus_expo_tiled"] = {"out": c2list(out2), "itern": it2, "layout": list(top2.tiled_layout())}
    top2.close()
    ctx.set_tuning("tl_force", None)
    ctx.set_tuning("pb_row_block", None)
    # --- matrix-free lattice operators: halo exchange instead of the all-gather
    #     (a) open 2-D Laplacian, (b) 3-D complex hops, periodic in every dimension (ring neighbours wrap around)
    side = 24
    n4 = side * side
    rb4, nl4 = ctx.partition(n4)
    st = L.StencilOperator(ctx, [side, side], diag=4.0, hop=-1.0, row_begin=rb4, n_local=nl4)
    e4 = L.LambdaLanczos(st, n4, False, 1)
    e4.eigenvalue_offset = -8.0
    e4.init_vector = lambda v, row_begin: np.copyto(v, G.start_vector(v.shape[0], 1, np.float64, row_begin))
    v4, x4 = e4.run()
    res["stencil_laplace"] = {"vals": v4.tolist(), "vecs": [x4[0].tolist()], "iters": e4.getIterationCounts(),
                              "alpha": e4.last_alpha.tolist()}
    st.close()
    dims = [9, 5, 7]
    n5 = int(np.prod(dims))
    rb5, nl5 = ctx.partition(n5)
    ons = 0.3 * np.cos(np.arange(rb5, rb5 + nl5))
    st3 = L.StencilOperator(ctx, dims, diag=0.25, hop=[0.5 + 1j, -1.0, 0.75j], periodic=True, onsite=ons,
                            dtype=np.complex128, row_begin=rb5, n_local=nl5)
    x5 = G.start_vector(nl5, 3, np.complex128, rb5)
    xd, yd = ctx.to_device(x5), ctx.empty(nl5, np.complex128)
    dot5 = L.spmv(st3, xd, yd, offset=-0.5, want_dot=True)
    out5, it5 = L.Exponentiator(st3, n5).run(-0.7j, x5)
    res["stencil_3d"] = {"y": c2list(yd.get()), "dot": dot5, "out": c2list(out5), "itern": it5}
    st3.close()
    # (c) the same with shard boundaries and a fastest dimension that are multiples of 8: the vectorised lattice kernel
    dims_v = [12, 6, 8]
    n7 = int(np.prod(dims_v))
    rb7, nl7 = ctx.partition(n7)
    stv = L.StencilOperator(ctx, dims_v, diag=0.25, hop=[0.5 + 1j, -1.0, 0.75j], periodic=[True, False, True],
                            onsite=0.3 * np.cos(np.arange(rb7, rb7 + nl7)), dtype=np.complex128, row_begin=rb7, n_local=nl7,
                            phase_grad=[[0.1, 0.2, 0.3], [0.0, 0.4, -0.2], [0.5, 0.0, 0.7]])
    x7 = G.start_vector(nl7, 5, np.complex128, rb7)
    xd, yd = ctx.to_device(x7), ctx.empty(nl7, np.complex128)
    dot7 = L.spmv(stv, xd, yd, offset=0.25, want_dot=True)
    res["stencil_vec"] = {"y": c2list(yd.get()), "dot": dot7}
    stv.close()
    # --- dense row block (all-gather path)
    n6 = 203
    rb6, nl6 = ctx.partition(n6)
    rng = np.random.default_rng(5)
    a6 = rng.standard_normal((n6, n6))
    a6 = a6 + a6.T
    dn = L.DenseOperator(ctx, a6[rb6:rb6 + nl6], row_begin=rb6)
    e6 = L.LambdaLanczos(dn, n6, True, 1)
    e6.init_vector = lambda v, row_begin: np.copyto(v, G.start_vector(v.shape[0], 1, np.float64, row_begin))
    v6, x6 = e6.run()
    res["dense"] = {"vals": v6.tolist(), "vecs": [x6[0].tolist()], "iters": e6.getIterationCounts()}
    dn.close()
    # the same with shard boundaries on 16-byte pieces of the rows (n = 208: shards of 104 / 70 / 52 rows): the column-split form
    # (own columns under the all-gather, the rest after it) with vector loads; n = 203 above takes gather-then-multiply
    n9 = 208
    rb9, nl9 = ctx.partition(n9)
    a9 = rng.standard_normal((n9, n9))
    a9 = a9 + a9.T
    dn9 = L.DenseOperator(ctx, a9[rb9:rb9 + nl9], row_begin=rb9)
    e9 = L.LambdaLanczos(dn9, n9, True, 1)
    e9.init_vector = lambda v, row_begin: np.copyto(v, G.start_vector(v.shape[0], 1, np.float64, row_begin))
    v9, x9 = e9.run()
    res["dense_aligned"] = {"vals": v9.tolist(), "vecs": [x9[0].tolist()], "iters": e9.getIterationCounts()}
    dn9.close()
    # --- banded random matrix (the band wraps around: first and last rank need each other's columns) through the 2-D tiled
    #     kernel: the row blocks whose tiles are all own-column tiles run under the all-gather, the others behind it
    n10, band = 30011, 700
    rb10, nl10 = ctx.partition(n10)
    ctx.set_tuning("tl_force", "1")          # (a matrix this small is not eligible by itself: 16 KiB tiles)
    ctx.set_tuning("pb_row_block", "96")
    csr10 = G.randsym(n10, band=band, row_begin=rb10, n_local=nl10)
    x10 = G.start_vector(nl10, 7, np.float64, rb10)
    res["tiled"] = {}
    for acc, label in ((L.capi.ACCURACY_NORMWISE, "fixed"), (L.capi.ACCURACY_COMPONENTWISE, "ordered")):
        op = L.CsrOperator(ctx, *csr10, n_cols=n10, row_begin=rb10, accuracy=acc, kernel=L.capi.SPMV_TILED)
        assert op.selected_spmv() == L.capi.SPMV_TILED
        xd, yd = ctx.to_device(x10), ctx.empty(nl10)
        dot = L.spmv(op, xd, yd, offset=-0.25, want_dot=True)
        rec = {"y": yd.get().tolist(), "dot": dot, "layout": list(op.tiled_layout()), "n_local": nl10}
        if label == "fixed":
            eng = L.LambdaLanczos(op, n10, True, 2)     # two roots: a restart pass with a locked, sharded eigenvector
            eng.max_iteration = 40
            eng.init_vector = lambda v, row_begin: np.copyto(v, G.start_vector(v.shape[0], 1, np.float64, row_begin))
            vals, vecs = eng.run()
            rec.update(vals=vals.tolist(), alpha=eng.last_alpha.tolist(), iters=eng.getIterationCounts(), vecs=[v.tolist() for v in vecs])
        res["tiled"][label] = rec
        op.close()
    ctx.set_tuning("tl_force", None)
    ctx.set_tuning("pb_row_block", None)
    # --- fewer rows than ranks: the last shard(s) are empty
    tiny = np.array([[2.0, 1.0], [1.0, 3.0]])
    rb8, nl8 = ctx.partition(2)
    rp8 = np.arange(nl8 + 1, dtype=np.int64) * 2
    t_op = L.CsrOperator(ctx, rp8, np.tile(np.arange(2, dtype=np.int32), nl8), tiny[rb8:rb8 + nl8].reshape(-1), n_cols=2,
                         row_begin=rb8)
    e8 = L.LambdaLanczos(t_op, 2, True, 1)
    e8.init_vector = lambda v, row_begin: np.copyto(v, np.array([1.0, 0.5])[row_begin:row_begin + v.shape[0]])
    v8, x8 = e8.run()
    res["tiny"] = {"vals": v8.tolist(), "vecs": [x8[0].tolist()], "n_local": nl8}
    t_op.close()
    with open(os.path.join(out_dir, "rank%d.json" % rank), "w") as f:
        json.dump(res, f)
    ctx.close()


if __name__ == "__main__":
    main()
