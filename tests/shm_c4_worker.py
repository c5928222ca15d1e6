"""One rank of BASELINE config 4 at FULL SIZE on the test box's single GPU: the n = 1e7, nnz = 1.5e8 random symmetric matrix
row-partitioned over `world` ranks (ll_partition), every rank a process of its own with the production sharded path — PB image
with own / remote column-block ranges, chunk-major gather buffer of (P + 1) * n_shard elements, overlapped exchange, all-reduced
Gram-Schmidt columns, replicated host decisions — talking through the host-staged TEST transport (RCCL refuses several ranks on
one device).  tests/test_gpu_round2.py::test_c4_full_size_* starts `world` of these and checks their records.
argv: rank world shm_name out_dir n window"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import lambda_lanczos_amd as L  # noqa: E402
from util import install_hook_sync  # noqa: E402
from lambda_lanczos_amd import generators as G  # noqa: E402


install_hook_sync()   # the harness's hook settings (util.HOOK_KEYS in os.environ) -> every context of this process


def main():
    rank, world, name, out_dir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    n, window = int(sys.argv[5]), int(sys.argv[6])
    t0 = time.time()
    ctx = L.Context(0)
    ctx.init_comm(name.encode() + b"\0" * (128 - len(name)), rank, world)
    rb, nl = ctx.partition(n)
    csr = G.randsym(n, row_begin=rb, n_local=nl)
    init = G.start_vector_fast(nl, 1, np.float64, rb)
    op = L.CsrOperator(ctx, *csr, n_cols=n, row_begin=rb)
    nnz_local = int(csr[0][-1])
    del csr
    res = {"rank": rank, "row_begin": rb, "n_local": nl, "nnz_local": nnz_local, "kernel": op.selected_spmv(),
           "accuracy": op.accuracy(), "seconds_setup": time.time() - t0}

    def one_pass():
        xd, yd = ctx.to_device(init), ctx.empty(nl)
        dot = L.spmv(op, xd, yd, offset=0.5, want_dot=True)
        y = yd.get()
        xd.free()
        yd.free()
        eng = L.LambdaLanczos(op, n, True, 1)
        eng.max_iteration = window
        eng.init_vector = lambda v, row_begin: np.copyto(v, G.start_vector_fast(v.shape[0], 1, np.float64, row_begin))
        t1 = time.time()
        vals, vecs = eng.run()
        return dict(y=y, dot=dot, vals=vals.copy(), vec=vecs[0].copy(), alpha=eng.last_alpha.copy(), beta=eng.last_beta.copy(),
                    iters=eng.getIterationCounts(), lagged=int(eng.last_stats["lagged_iterations"]),
                    pair=int(eng.last_stats["pair_iterations"]), seconds=time.time() - t1)

    # overlapped exchange (production default): all-gather in chunks on the communication stream, own-column blocks under it
    os.environ["LL_COMM_OVERLAP"] = "1"
    ctx.reload_env()
    a = one_pass()
    # serial issue order (everything on one stream): must be the same bits
    os.environ["LL_COMM_OVERLAP"] = "0"
    ctx.reload_env()
    b = one_pass()
    res["serial_equals_overlapped"] = bool(
        np.array_equal(a["y"], b["y"]) and a["dot"] == b["dot"] and np.array_equal(a["vals"], b["vals"]) and
        np.array_equal(a["vec"], b["vec"]) and np.array_equal(a["alpha"], b["alpha"]) and np.array_equal(a["beta"], b["beta"]))
    res.update(dot=a["dot"], vals=a["vals"].tolist(), alpha=a["alpha"].tolist(), beta=a["beta"].tolist(), iters=a["iters"],
               lagged=a["lagged"], pair=a["pair"], seconds_run_overlapped=a["seconds"], seconds_run_serial=b["seconds"])
    np.save(os.path.join(out_dir, "y_rank%d.npy" % rank), a["y"])
    np.save(os.path.join(out_dir, "vec_rank%d.npy" % rank), a["vec"])
    op.close()
    with open(os.path.join(out_dir, "rank%d.json" % rank), "w") as f:
        json.dump(res, f)
    ctx.close()


if __name__ == "__main__":
    main()
