"""Worker of tests/test_dist_gloo.py: one process per rank under torch.distributed.run (gloo, CPU).

It executes the SAME distributed algorithm the HIP engine runs on a sharded context (lambda-lanczos_amd/csrc/
engine.cpp: Engine::apply / Engine::orth) — 1-D contiguous row partition from ll_partition, one all-gather of the
current Lanczos vector per SpMV (equal shard strides, padded last shard), all-reduce(sum) of alpha, of the k+1
block Gram-Schmidt coefficients (+ ||w||^2) and of the norm — with numpy standing in for the kernels and gloo for
RCCL.  The HIP kernels themselves cannot run without a GPU; what this pins is the sharding arithmetic and the
collective pattern (what each rank sends, receives and reduces, and that every rank takes the same decisions)."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import lambda_lanczos_amd as L  # noqa: E402
from util import install_hook_sync  # noqa: E402
from lambda_lanczos_amd import generators as G  # noqa: E402


install_hook_sync()   # the harness's hook settings (util.HOOK_KEYS in os.environ) -> every context of this process


def all_reduce(a):
    t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64).copy())
    dist.all_reduce(t)
    return t.numpy()


def all_gather_x(x_local, n, shard):
    """ncclAllGather semantics: every rank contributes exactly `shard` elements (padded), position = rank*shard."""
    world = dist.get_world_size()
    send = np.zeros(shard)
    send[: x_local.shape[0]] = x_local
    out = [torch.zeros(shard, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(out, torch.from_numpy(send))
    return np.concatenate([o.numpy() for o in out])[: max(n, 1)]


def sharded_lanczos(csr_local, n, rb, nl, init_local, find_max, max_iteration, eps, offset, nroot=5):
    import scipy.sparse as sp

    world = dist.get_world_size()
    shard = -(-n // world)
    a_loc = sp.csr_matrix((csr_local[2], csr_local[1], csr_local[0]), shape=(nl, n))
    u = [init_local / np.sqrt(all_reduce([init_local @ init_local])[0])]
    alpha, beta, pevs = [], [], None
    itern = max_iteration
    for k in range(1, max_iteration + 1):
        x_full = all_gather_x(u[k - 1], n, shard)                     # exchange step
        w = a_loc @ x_full + offset * u[k - 1]
        a_k = all_reduce([u[k - 1] @ w])[0]                           # alpha: one scalar
        alpha.append(a_k)
        w = w - a_k * u[k - 1] - (beta[-1] * u[k - 2] if k > 1 else 0.0)
        basis = np.array(u)
        red = all_reduce(np.concatenate([basis @ w, [w @ w]]))        # k coefficients + ||w||^2 in ONE all-reduce
        h, c0 = red[:-1], red[-1]
        w = w - h @ basis
        c1 = all_reduce([w @ w])[0]
        if c1 < 0.5 * c0:                                             # DGKS second pass, same test on every rank
            h2 = all_reduce(basis @ w)
            w = w - h2 @ basis
            c1 = all_reduce([w @ w])[0]
        beta.append(np.sqrt(c1))
        ev, _, _ = L.tridiag_eig(alpha, beta[:-1], want_vectors=True)  # host step, identical on every rank
        evs = (ev[::-1] if find_max else ev)[: min(nroot, k)]
        if beta[-1] < np.finfo(float).eps * 10:
            itern = k
            break
        u.append(w / beta[-1])
        if pevs is not None and len(pevs) == len(evs) and all(
                abs(e - p) < min(abs(e), abs(p)) * eps for e, p in zip(evs[:nroot], pevs[:nroot])):
            itern = k
            break
        pevs = evs
    m = len(alpha)
    ev, q, _ = L.tridiag_eig(alpha, beta[: m - 1])
    idx = m - 1 if find_max else 0
    vec = q[idx] @ np.array(u[:m])
    vec = vec / np.sqrt(all_reduce([vec @ vec])[0])
    return ev[idx] - offset, vec, itern, np.array(alpha), np.array(beta)


def main():
    out_dir = sys.argv[1]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    results = {}
    for name, n, gen, kw, find_max, offset, max_it in [
        ("randsym", 3001, G.randsym, {}, True, 0.0, 3001),
        ("laplace", 28 * 28, lambda n_, row_begin, n_local: G.laplace2d(28, row_begin, n_local), {}, False, -8.0, 784),
    ]:
        rb, nl = L.partition(n, world, rank)                          # the product's own partition function
        csr = gen(n, row_begin=rb, n_local=nl, **kw)
        init = G.start_vector(nl, 1, np.float64, rb)
        lam, vec, it, al, be = sharded_lanczos(csr, n, rb, nl, init, find_max, max_it, np.finfo(float).eps * 1e3, offset)
        results[name] = {"rank": rank, "row_begin": rb, "n_local": nl, "lambda": lam, "itern": it,
                         "vec": vec.tolist(), "alpha": al.tolist(), "beta": be.tolist(),
                         "nnz_local": int(csr[0][-1])}
    with open(os.path.join(out_dir, "rank%d.json" % rank), "w") as f:
        json.dump(results, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
