"""Seeded randomised sweep of the whole-loop entry points against the oracle: sizes from 1 up, ragged sparse symmetric
and Hermitian matrices (empty rows, isolated 1x1 blocks => invariant subspaces / breakdown), both ends of the spectrum,
several roots, offsets, every orthogonalisation and tridiagonal mode, both SpMV kernels."""
import numpy as np
import pytest

import lambda_lanczos_amd as L
from util import overlap

pytestmark = pytest.mark.gpu


def random_hermitian(rng, n, density, complex_):
    import scipy.sparse as sp

    m = sp.random(n, n, density=density, random_state=np.random.RandomState(rng.integers(1 << 30)), format="csr")
    m.data = rng.uniform(-1, 1, m.data.shape)
    if complex_:
        m = m.astype(np.complex128)
        m.data = m.data + 1j * rng.uniform(-1, 1, m.data.shape)
    a = (m + m.getH()).tocsr()
    a.setdiag(rng.uniform(-2, 2, n))
    a = a.tocsr()
    a.sort_indices()
    # cut a few rows/columns out completely (empty rows: isolated zero eigenvalues)
    return a.indptr.astype(np.int64), a.indices.astype(np.int32), a.data.copy(), a


import os

CASES = list(range(36))
# one-off stress runs: LL_FUZZ_SEEDS=a:b adds the seeds a .. b-1 (seeds from 36 up also draw the 2-D tiled kernel, forced onto matrices
# that are not eligible for it by themselves); the suite the driver runs keeps the 36 cases above
if os.environ.get("LL_FUZZ_SEEDS"):
    _a, _b = os.environ["LL_FUZZ_SEEDS"].split(":")
    CASES = CASES + [c for c in range(int(_a), int(_b)) if c >= 36]
    EXTRA = [c for c in range(int(_a), int(_b)) if c >= 36]
else:
    EXTRA = []


# geometry "streaming": LL_BLAS_SMALL_BYTES=0 puts these small problems on the streaming kernels, i.e. (orth mode 0) on the
# one-sweep Gram-Schmidt form with its repair paths: breakdown, invariant subspaces, locked eigenvectors of restart passes
@pytest.mark.parametrize("geometry", ["default", "streaming"])
@pytest.mark.parametrize("seed", CASES)
def test_random_problem_matches_oracle(ctx, oracle, seed, geometry, llenv):
    if geometry == "streaming":
        llenv.setenv("LL_BLAS_SMALL_BYTES", "0")
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.choice([1, 2, 3, 5, 17, 64, 200, 777, 2500]))
    complex_ = bool(rng.integers(2))
    density = float(rng.choice([0.02, 0.1, 0.5])) if n > 5 else 1.0
    rp, ci, va, a = random_hermitian(rng, n, density, complex_)
    find_max = bool(rng.integers(2))
    k = int(min(n, rng.choice([1, 1, 2, 4])))
    spread = float(np.abs(va).sum() / max(n, 1) + 3.0)
    offset = float(rng.choice([0.0, spread, -spread])) if not find_max else float(rng.choice([0.0, spread]))
    if not find_max and offset >= 0:
        offset = -spread          # make the wanted (smallest) Ritz values large in magnitude
    dtype = np.complex128 if complex_ else np.float64
    init = rng.uniform(-1, 1, n).astype(dtype)
    if complex_:
        init = init + 1j * rng.uniform(-1, 1, n)
    if seed < 36:
        llenv.setenv("LL_SPMV_KERNEL", ("csr", "pb")[int(rng.integers(2))])
    else:
        kern = ("csr", "pb", "tiled")[int(rng.integers(3))]
        if kern == "tiled" and len(va) == 0:
            kern = "pb"          # (the tiled kernel by name is an error for a matrix without entries)
        if kern == "tiled":
            llenv.setenv("LL_TL_FORCE", "1")
        llenv.setenv("LL_SPMV_KERNEL", kern)
    op = L.CsrOperator(ctx, rp, ci, va)
    eng = L.LambdaLanczos(op, n, find_max, k)
    eng.eigenvalue_offset = offset
    eng.orth_mode = int(rng.integers(3))
    eng.tridiag_mode = int(rng.choice([L.TRIDIAG_QR, L.TRIDIAG_AUTO]))
    eng.init_vector = lambda v, *_: np.copyto(v, init)
    vals, vecs = eng.run()
    ora = oracle.lanczos((rp, ci, va), init, find_max, num_eigs=k, offset=offset)
    assert len(vals) == len(ora["eigenvalues"])
    scale = max(1.0, float(np.max(np.abs(ora["eigenvalues"] + offset))))
    assert np.max(np.abs(vals - ora["eigenvalues"])) <= 1e-9 * scale, (seed, vals, ora["eigenvalues"])
    dense = a.toarray()
    w = np.linalg.eigvalsh(dense)
    for i, lam in enumerate(vals):
        # a true eigenpair of A (degenerate eigenvalues may pick another vector of the eigenspace than the oracle)
        r = np.linalg.norm(dense @ vecs[i] - lam * vecs[i])
        assert r <= 1e-6 * scale, (seed, i, r)
        assert np.min(np.abs(w - lam)) <= 1e-8 * scale
        assert abs(np.linalg.norm(vecs[i]) - 1) <= 1e-10
    if k == 1 and len(w) > 1:
        gap = np.min(np.abs(np.delete(w, np.argmin(np.abs(w - vals[0]))) - vals[0]))
        if gap > 1e-6 * scale:
            assert 1 - overlap(vecs[0], ora["eigenvectors"][0]) <= 1e-7
    op.close()


@pytest.mark.parametrize("seed", list(range(8)) + EXTRA)
def test_random_exponentiator_matches_oracle(ctx, oracle, seed):
    rng = np.random.default_rng(2000 + seed)
    n = int(rng.choice([1, 2, 9, 120, 1500]))
    complex_ = bool(rng.integers(2))
    rp, ci, va, a = random_hermitian(rng, n, 0.2 if n > 9 else 1.0, complex_)
    dtype = np.complex128 if complex_ else np.float64
    inp = rng.uniform(-1, 1, n).astype(dtype)
    a_coef = (1j * rng.uniform(-2, 2)) if complex_ else float(rng.uniform(-1.5, 1.5))
    op = L.CsrOperator(ctx, rp, ci, va)
    ex = L.Exponentiator(op, n)
    ex.full_orthogonalize = bool(rng.integers(2))
    kw = {}
    if seed >= 8:   # the extra seeds of a stress run: a real exponent never meets the reference's stop test (EX:154) and would run n
        ex.max_iteration = min(n, 60)   # iterations with an O(k^3) host step each — bound the run on both sides
        kw["max_iteration"] = min(n, 60)
    out, it = ex.run(a_coef, inp)
    o_out, o_it, _ = oracle.expo((rp, ci, va), a_coef, inp, full_orthogonalize=ex.full_orthogonalize, **kw)
    assert abs(it - o_it) <= 1
    if it == o_it:
        assert np.linalg.norm(out - o_out) <= 1e-8 * max(np.linalg.norm(o_out), 1e-300)
    else:
        # The reference stops when SUCCESSIVE approximations overlap to eps = 1e2 * epsilon (EX:147-158) — a test that is quadratic in
        # their difference, so two successive iterates differ by up to sqrt(2 eps) ~ 2e-7 of their norm; where rounding puts the stop one
        # iteration apart (stress seed 239: 38 against 37 iterations, 6e-8 apart, each within 6e-8 of the exact exponential) the two
        # outputs are held to the reference's own criterion.
        ov = abs(np.vdot(out, o_out)) / (np.linalg.norm(out) * np.linalg.norm(o_out))
        assert 1 - ov <= 10 * ex.eps
    wv, v = np.linalg.eigh(a.toarray())
    exact = v @ (np.exp(a_coef * wv) * (v.conj().T @ inp))
    if "max_iteration" not in kw or it < kw["max_iteration"]:   # (a run cut off by the bound of the stress seeds is compared with the oracle only)
        assert np.linalg.norm(out - exact) <= 1e-6 * np.linalg.norm(exact)
    op.close()


@pytest.mark.parametrize("dtype,side", [(np.float64, 41), (np.complex128, 29)])
def test_full_krylov_space_many_basis_groups(ctx, oracle, dtype, side):
    """Runs the loop until the Krylov space is (numerically) exhausted: k grows past the per-launch basis limit of the
    multi-dot kernel (1535 real / 767 complex vectors), so the multi-group orthogonalisation path, slab growth and the
    near-breakdown end of the iteration are exercised; the answer is checked against dense eigenvalues."""
    from lambda_lanczos_amd import generators as G

    csr = G.laplace2d_np(side) if dtype == np.float64 else G.torus_np(side)
    n = side * side
    init = G.start_vector(n, 5, dtype)
    op = L.CsrOperator(ctx, *csr)
    eng = L.LambdaLanczos(op, n, False, 1)
    eng.eps = 1e-30                      # the relative stop test can never fire: run to max_iteration / breakdown
    eng.eigenvalue_offset = -9.0
    eng.init_vector = lambda v, *_: np.copyto(v, init)
    vals, vecs = eng.run()
    its = eng.getIterationCounts()[0]
    assert its > (1535 if dtype == np.float64 else 767)
    import scipy.sparse as sp

    dense = sp.csr_matrix((csr[2], csr[1], csr[0]), shape=(n, n)).toarray()
    w = np.linalg.eigvalsh(dense)
    assert abs(vals[0] - w[0]) <= 1e-10 * 9
    assert np.linalg.norm(dense @ vecs[0] - vals[0] * vecs[0]) <= 1e-8 * 9
    op.close()


@pytest.mark.parametrize("seed", list(range(24)) + EXTRA)
def test_random_lattice_operator_matches_oracle(ctx, oracle, seed):
    """Random lattices (1-3 dimensions, open/periodic mix, lengths from 1 up, complex hops, on-site terms, all four
    storage types): one apply of the matrix-free operator against the oracle's CSR row loop on the equivalent matrix,
    and a short Exponentiator run through it."""
    from lambda_lanczos_amd import generators as G

    rng = np.random.default_rng(3000 + seed)
    nd = int(rng.integers(1, 4))
    dims = [int(rng.choice([1, 2, 3, 5, 8, 13, 31])) for _ in range(nd)]
    dtype = [np.float64, np.complex128, np.float32, np.complex64][seed % 4]
    cplx = np.issubdtype(np.dtype(dtype), np.complexfloating)
    single = np.dtype(dtype).itemsize == (8 if cplx else 4)
    wide = np.complex128 if cplx else np.float64
    hop = rng.uniform(-1, 1, nd) + (1j * rng.uniform(-1, 1, nd) if cplx else 0)
    periodic = [bool(b) for b in rng.integers(0, 2, nd)]
    n = int(np.prod(dims))
    onsite = rng.uniform(-1, 1, n) if rng.integers(2) else None
    if onsite is not None and single:
        onsite = onsite.astype(np.float32).astype(np.float64)
    hop_dev = hop.astype(np.complex64).astype(np.complex128) if single else hop   # what a float operator can hold
    diag = float(np.float32(rng.uniform(-2, 2)))
    kw = dict(diag=diag, hop=list(hop_dev), periodic=periodic, onsite=onsite)
    csr = G.lattice_csr(dims, dtype=wide, **kw)
    op = L.StencilOperator(ctx, dims, dtype=dtype, **kw)
    x = (rng.uniform(-1, 1, n) + (1j * rng.uniform(-1, 1, n) if cplx else 0)).astype(dtype)
    xd, yd = ctx.to_device(x), ctx.empty(n, dtype)
    alpha = L.spmv(op, xd, yd, offset=0.5, want_dot=True)
    xw = x.astype(wide)
    y_ref = oracle.spmv(csr, xw) + 0.5 * xw
    tol = 32 * (np.finfo(np.float32).eps if single else np.finfo(np.float64).eps)
    scale = op.inf_norm() + 1.5
    assert np.max(np.abs(yd.get() - y_ref)) <= tol * scale, (seed, dims, periodic)
    assert abs(alpha - np.vdot(xw, y_ref).real) <= tol * scale * n
    if not single:
        a = -0.3j if cplx else -0.3
        ex = L.Exponentiator(op, n)
        ex.max_iteration = min(n, 40)   # a real exponent never meets the reference's stop test (EX:154): bound the run
        out, it = ex.run(a, x)
        o_ref, it_ref, _ = oracle.expo(csr, a, xw, max_iteration=min(n, 40))
        assert abs(it - it_ref) <= 1, (seed, it, it_ref)
        assert np.max(np.abs(out - o_ref)) <= 1e-10 * max(1.0, np.linalg.norm(xw))
    op.close()
