"""Synthetic inputs of SURVEY.md section 8(d): deterministic functions of (index, seed) through splitmix64.

Two implementations that must agree bit for bit: numpy (this file; any size that fits numpy, used for the small
parity cases) and C++ (csrc/generators.cpp -> lib/libllgen.so; used at BASELINE sizes, n = 1e7 in a few seconds).
This is workload synthesis for bench.py / tests — neither the hot path nor the oracle.
"""
import ctypes as C
import math
import os

# the generator library's OpenMP workers must go to sleep right after a parallel region: workers that keep spinning
# (libomp: 200 ms) disturb launch-bound GPU runs that follow immediately (tools/stall_probe.py: one ~85 ms stall per
# process).  Read by the OpenMP runtimes when they start, so it is set before the library is loaded.
os.environ.setdefault("OMP_WAIT_POLICY", "passive")

import numpy as np  # noqa: E402

from . import _capi as capi

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
K_OUT = 7


def splitmix64(x):
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def u01(x):
    return (splitmix64(x) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def start_vector(n_local, seed=1, dtype=np.float64, row_begin=0):
    """v_i = 2*u01(seed*2^40 + i) - 1 (complex: re from 2i, im from 2i+1), i = global index."""
    base = np.uint64(seed) << np.uint64(40)
    g = np.arange(row_begin, row_begin + n_local, dtype=np.uint64)
    if np.dtype(dtype) == np.complex128:
        re = 2.0 * u01(base + np.uint64(2) * g) - 1.0
        im = 2.0 * u01(base + np.uint64(2) * g + np.uint64(1)) - 1.0
        return re + 1j * im
    return 2.0 * u01(base + g) - 1.0


# ------------------------------------------------------------------ numpy versions (small n)
def laplace2d_np(N, row_begin=0, n_local=None):
    n = N * N
    n_local = n - row_begin if n_local is None else n_local
    r = np.arange(row_begin, row_begin + n_local, dtype=np.int64)
    y, x = r // N, r % N
    cols = np.stack([r - N, r - 1, r, r + 1, r + N], axis=1)
    vals = np.tile(np.array([-1.0, -1.0, 4.0, -1.0, -1.0]), (n_local, 1))
    mask = np.stack([y > 0, x > 0, np.ones_like(r, bool), x + 1 < N, y + 1 < N], axis=1)
    rp = np.concatenate([[0], np.cumsum(mask.sum(axis=1))]).astype(np.int64)
    return rp, cols[mask].astype(np.int32), vals[mask]


def _b_cols_np(n, band):
    i = np.arange(n, dtype=np.int64)[:, None]
    j = np.arange(K_OUT, dtype=np.int64)[None, :]
    h = splitmix64((64 * i + j).astype(np.uint64))
    if band <= 0:
        c = (h % np.uint64(n - 1)).astype(np.int64)
        c = c + (c >= i)
    else:
        off = (h % np.uint64(2 * band)).astype(np.int64)
        d = off - band
        d = d + (d >= 0)
        c = (i + d) % n
    v = 2.0 * u01((64 * i + j + 32).astype(np.uint64)) - 1.0
    return c, v


def randsym_np(n, band=0, row_begin=0, n_local=None):
    """A = B + B^T + 7I, 7 random out-entries per row of B, duplicates kept; rows sorted by column (stable)."""
    n_local = n - row_begin if n_local is None else n_local
    c, v = _b_cols_np(n, band)
    i = np.repeat(np.arange(n, dtype=np.int64), K_OUT)
    cf, vf = c.reshape(-1), v.reshape(-1)
    # (row, col, val, order-key): own entries first (by j), then diagonal, then transposed entries (by source row, j)
    rows = np.concatenate([i, np.arange(n, dtype=np.int64), cf])
    cols = np.concatenate([cf, np.arange(n, dtype=np.int64), i])
    vals = np.concatenate([vf, np.full(n, 7.0), vf])
    keep = (rows >= row_begin) & (rows < row_begin + n_local)
    rows, cols, vals = rows[keep], cols[keep], vals[keep]
    order = np.lexsort((np.arange(rows.shape[0]), cols, rows))  # stable within equal (row, col)
    rows, cols, vals = rows[order], cols[order], vals[order]
    rp = np.zeros(n_local + 1, dtype=np.int64)
    np.add.at(rp, rows - row_begin + 1, 1)
    return np.cumsum(rp), cols.astype(np.int32), vals


def torus_np(N, row_begin=0, n_local=None):
    n = N * N
    n_local = n - row_begin if n_local is None else n_local
    phi = 2.0 * math.pi * 3.0 / N
    r = np.arange(row_begin, row_begin + n_local, dtype=np.int64)
    y, x = r // N, r % N
    cols = np.stack([y * N + (x + 1) % N, y * N + (x + N - 1) % N, ((y + 1) % N) * N + x, ((y + N - 1) % N) * N + x, r],
                    axis=1)
    ph = phi * y.astype(np.float64)
    vals = np.stack([-np.cos(ph) - 1j * np.sin(ph), -np.cos(ph) + 1j * np.sin(ph), np.full(n_local, -1.0 + 0j),
                     np.full(n_local, -1.0 + 0j), (u01(r.astype(np.uint64)) - 0.5) + 0j], axis=1)
    order = np.argsort(cols, axis=1, kind="stable")
    cols = np.take_along_axis(cols, order, axis=1)
    vals = np.take_along_axis(vals, order, axis=1)
    rp = (5 * np.arange(n_local + 1)).astype(np.int64)
    return rp, cols.reshape(-1).astype(np.int32), vals.reshape(-1)


def dense_to_csr(a):
    """All entries of a small dense matrix as CSR (zeros included, like the reference's dense test lambdas)."""
    a = np.asarray(a)
    n = a.shape[0]
    rp = (n * np.arange(n + 1)).astype(np.int64)
    ci = np.tile(np.arange(n, dtype=np.int32), n)
    return rp, ci, np.ascontiguousarray(a.reshape(-1))


def coo_to_csr(n, rows, cols, vals):
    """{r, c, value} triplets (sample2_sparse.cpp:14-47) to CSR, stable in input order within a row."""
    rows = np.asarray(rows, dtype=np.int64)
    order = np.argsort(rows, kind="stable")
    rp = np.zeros(n + 1, dtype=np.int64)
    np.add.at(rp, rows + 1, 1)
    return np.cumsum(rp), np.asarray(cols, dtype=np.int32)[order], np.asarray(vals)[order]


def ring_csr(n, t=-1.0, dtype=np.float64):
    """Periodic 1-D chain with hopping t (T1:493-501, T2:113-121)."""
    r = np.arange(n)
    cols = np.stack([(r - 1) % n, (r + 1) % n], axis=1)
    cols.sort(axis=1)
    return (2 * np.arange(n + 1)).astype(np.int64), cols.reshape(-1).astype(np.int32), np.full(2 * n, t, dtype=dtype)


def chain_csr(n, t=-1.0):
    """Open 1-D chain (sample3_dynamic.cpp:17-22, T1:265-273)."""
    rows, cols = [], []
    for i in range(n - 1):
        rows += [i, i + 1]
        cols += [i + 1, i]
    return coo_to_csr(n, rows, cols, np.full(len(rows), t))


def lattice_csr(dims, diag=0.0, hop=-1.0, periodic=False, onsite=None, dtype=np.float64, row_begin=0, n_local=None,
                phase_grad=None):
    """CSR image of the matrix-free lattice operator (ll_op_create_stencil_*): rows [row_begin, row_begin+n_local),
    entries in the operator's own order (lower neighbours slowest dimension first, diagonal, upper neighbours fastest
    first); a neighbour reached twice (periodic dimension of length 1 or 2) appears twice."""
    dims = [int(d) for d in np.atleast_1d(dims)]
    nd = len(dims)
    hop = np.broadcast_to(np.asarray(hop, dtype=np.complex128), (nd,))
    periodic = np.broadcast_to(np.asarray(periodic, dtype=bool), (nd,))
    n = int(np.prod(dims))
    n_local = n - row_begin if n_local is None else n_local
    r = np.arange(row_begin, row_begin + n_local, dtype=np.int64)
    coords = list(np.unravel_index(r, dims))
    strides = [int(np.prod(dims[k + 1:])) for k in range(nd)]
    cols, vals, have = [], [], []

    pg = None if phase_grad is None else np.asarray(phase_grad, dtype=np.float64).reshape(nd, nd)

    def neighbour(k, sign):
        c = coords[k] + sign
        ok = (c >= 0) & (c < dims[k])
        if periodic[k]:
            c, ok = c % dims[k], np.ones_like(ok)
        cols.append(r + (c - coords[k]) * strides[k])
        t = np.full(n_local, hop[k])
        if pg is not None and np.any(pg[k] != 0):   # Peierls phase taken at the bond's LOWER site
            low = [cc.copy() for cc in coords]
            if sign < 0:
                low[k] = c
            t = t * np.exp(1j * sum(pg[k, e] * low[e] for e in range(nd)))
        vals.append(np.conj(t) if sign < 0 else t)
        have.append(ok)

    for k in range(nd):
        neighbour(k, -1)
    cols.append(r.copy())
    vals.append(np.full(n_local, diag, dtype=np.complex128) + (0 if onsite is None else np.asarray(onsite)))
    have.append(np.ones(n_local, dtype=bool))
    for k in range(nd - 1, -1, -1):
        neighbour(k, +1)
    cols, vals, have = np.stack(cols, 1), np.stack(vals, 1), np.stack(have, 1)
    rp = np.concatenate([[0], np.cumsum(have.sum(1))]).astype(np.int64)
    va = vals[have]
    if not np.issubdtype(np.dtype(dtype), np.complexfloating):
        va = va.real
    return rp, cols[have].astype(np.int32), np.ascontiguousarray(va.astype(dtype))


# ------------------------------------------------------------------ C++ versions (BASELINE sizes)
_gen = None


def _lib():
    global _gen
    if _gen is None:
        if not os.path.exists(capi.GEN_PATH):
            raise RuntimeError("%s not built (run __graft_entry__.build())" % capi.GEN_PATH)
        g = C.CDLL(capi.GEN_PATH)
        i64, vp = C.c_int64, C.c_void_p
        g.llgen_splitmix64.restype, g.llgen_splitmix64.argtypes = C.c_uint64, [C.c_uint64]
        g.llgen_start_vector_d.argtypes = [C.c_uint64, i64, i64, vp]
        g.llgen_start_vector_z.argtypes = [C.c_uint64, i64, i64, vp]
        g.llgen_laplace2d_count.restype, g.llgen_laplace2d_count.argtypes = i64, [i64, i64, i64]
        g.llgen_laplace2d_fill.argtypes = [i64, i64, i64, vp, vp, vp]
        g.llgen_randsym_count.restype, g.llgen_randsym_count.argtypes = i64, [i64, i64, i64, i64]
        g.llgen_randsym_fill.argtypes = [i64, i64, i64, i64, vp, vp, vp]
        g.llgen_torus_fill.argtypes = [i64, i64, i64, vp, vp, vp]
        _gen = g
    return _gen


def start_vector_fast(n_local, seed=1, dtype=np.float64, row_begin=0):
    v = np.empty(n_local, dtype=dtype)
    fn = _lib().llgen_start_vector_z if np.dtype(dtype) == np.complex128 else _lib().llgen_start_vector_d
    fn(seed, row_begin, n_local, capi.ptr(v))
    return v


def laplace2d(N, row_begin=0, n_local=None):
    n_local = N * N - row_begin if n_local is None else n_local
    nnz = _lib().llgen_laplace2d_count(N, row_begin, n_local)
    rp, ci, va = np.empty(n_local + 1, np.int64), np.empty(nnz, np.int32), np.empty(nnz, np.float64)
    _lib().llgen_laplace2d_fill(N, row_begin, n_local, capi.ptr(rp), capi.ptr(ci), capi.ptr(va))
    return rp, ci, va


def randsym(n, band=0, row_begin=0, n_local=None):
    n_local = n - row_begin if n_local is None else n_local
    nnz = _lib().llgen_randsym_count(n, band, row_begin, n_local)
    rp, ci, va = np.empty(n_local + 1, np.int64), np.empty(nnz, np.int32), np.empty(nnz, np.float64)
    _lib().llgen_randsym_fill(n, band, row_begin, n_local, capi.ptr(rp), capi.ptr(ci), capi.ptr(va))
    return rp, ci, va


def torus(N, row_begin=0, n_local=None):
    n_local = N * N - row_begin if n_local is None else n_local
    rp, ci, va = np.empty(n_local + 1, np.int64), np.empty(5 * n_local, np.int32), np.empty(5 * n_local, np.complex128)
    _lib().llgen_torus_fill(N, row_begin, n_local, capi.ptr(rp), capi.ptr(ci), capi.ptr(va))
    return rp, ci, va


def laplace2d_lambda_min(N):
    """Analytic smallest eigenvalue of the N x N Dirichlet 5-point Laplacian."""
    return 4.0 - 4.0 * math.cos(math.pi / (N + 1))
