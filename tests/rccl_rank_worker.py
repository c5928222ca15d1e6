"""One rank of the 2-device RCCL test (tests/test_gpu_rccl2.py): rank r drives device r (or device 0 for every rank when
LL_TEST_SAME_DEVICE=1, together with the host-staged test transport in LL_COMM_PLUGIN — the harness check that runs on the
pool's 1-GPU boxes).  Usage: rccl_rank_worker.py RANK WORLD UNIQUE_ID_HEX OUT_DIR"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import lambda_lanczos_amd as L  # noqa: E402
from util import install_hook_sync  # noqa: E402
from lambda_lanczos_amd import generators as G  # noqa: E402


install_hook_sync()   # the harness's hook settings (util.HOOK_KEYS in os.environ) -> every context of this process


def main():
    rank, world, uid_hex, out_dir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    same = os.environ.get("LL_TEST_SAME_DEVICE") == "1"
    ctx = L.Context(0 if same else rank)
    uid = bytes.fromhex(uid_hex)
    ctx.init_comm(uid + b"\0" * (128 - len(uid)), rank, world)
    res = {"ranks_seen": ctx.ranks_seen()}
    n = 60013
    rb, nl = ctx.partition(n)
    csr = G.randsym(n, row_begin=rb, n_local=nl)
    x = G.start_vector(nl, 3, np.float64, rb)
    for label in ("pb", "csr"):
        os.environ["LL_SPMV_KERNEL"] = label
        ctx.reload_env()
        op = L.CsrOperator(ctx, *csr, n_cols=n, row_begin=rb)
        xd, yd = ctx.to_device(x), ctx.empty(nl)
        dot = L.spmv(op, xd, yd, offset=0.5, want_dot=True)
        res["spmv_" + label] = {"y": yd.get().tolist(), "dot": dot, "row_begin": rb}
        eng = L.LambdaLanczos(op, n, True, 2)
        eng.max_iteration = 60
        eng.init_vector = lambda v, row_begin: np.copyto(v, G.start_vector(v.shape[0], 1, np.float64, row_begin))
        vals, vecs = eng.run()
        res["lanczos_" + label] = {"vals": vals.tolist(), "vecs": [v.tolist() for v in vecs], "iters": eng.getIterationCounts(),
                                   "alpha": eng.last_alpha.tolist(), "row_begin": rb}
        op.close()
    # complex torus, Exponentiator with sharded input / output
    N = 40
    n3 = N * N
    rb3, nl3 = ctx.partition(n3)
    os.environ["LL_SPMV_KERNEL"] = "pb"
    ctx.reload_env()
    top = L.CsrOperator(ctx, *G.torus(N, rb3, nl3), n_cols=n3, row_begin=rb3)
    inp = G.start_vector(nl3, 1, np.complex128, rb3)
    out, it = L.Exponentiator(top, n3).run(-1j, inp)
    res["torus_expo"] = {"re": out.real.tolist(), "im": out.imag.tolist(), "itern": it}
    top.close()
    # lattice operator: ring halo exchange (ncclSend / ncclRecv)
    side = 48
    n4 = side * side
    rb4, nl4 = ctx.partition(n4)
    st = L.StencilOperator(ctx, [side, side], diag=4.0, hop=-1.0, row_begin=rb4, n_local=nl4)
    e4 = L.LambdaLanczos(st, n4, False, 1)
    e4.eigenvalue_offset = -8.0
    e4.max_iteration = 80
    e4.init_vector = lambda v, row_begin: np.copyto(v, G.start_vector(v.shape[0], 1, np.float64, row_begin))
    v4, x4 = e4.run()
    res["stencil"] = {"vals": v4.tolist(), "alpha": e4.last_alpha.tolist()}
    st.close()
    with open(os.path.join(out_dir, "rank%d.json" % rank), "w") as f:
        json.dump(res, f)
    ctx.close()


if __name__ == "__main__":
    main()
