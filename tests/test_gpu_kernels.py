"""Parity of every HIP kernel family against the CPU oracle, through the C ABI.

Floating point, so tolerance based: the oracle sums left to right, the kernels sum in wavefront trees.
Tolerances are stated next to each assertion as a multiple of eps * (problem scale)."""
import numpy as np
import pytest

import lambda_lanczos_amd as L
from lambda_lanczos_amd import generators as G

pytestmark = pytest.mark.gpu
EPS = np.finfo(np.float64).eps


def rnd(n, dtype, seed):
    return G.start_vector(n, seed, dtype)


def csr_cases():
    cases = {
        "dense3": G.dense_to_csr(np.array([[2.0, 1, 1], [1, 2, 1], [1, 1, 2]])),
        "laplace37": G.laplace2d_np(37),
        "randsym5000": G.randsym_np(5000),
        "banded5000": G.randsym_np(5000, band=64),
        "torus24": G.torus_np(24),
        "ring_complex": G.ring_csr(100, -1.0, np.complex128),
        "banded20000": G.randsym_np(20000, band=300),   # ten column tiles of the tiled kernel, two or three per row block
    }
    # ragged: empty rows, a row longer than one LDS tile (>1024 nnz), single-entry rows
    n = 3000
    rng = np.random.default_rng(7)
    lens = rng.integers(0, 9, size=n)
    lens[::17] = 0
    lens[1234] = 2500
    lens[2999] = 1500
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    ci = rng.integers(0, n, size=rp[-1]).astype(np.int32)
    cases["ragged"] = (rp, ci, rng.uniform(-1, 1, size=rp[-1]))
    cases["ragged_z"] = (rp, ci, rng.uniform(-1, 1, size=rp[-1]) + 1j * rng.uniform(-1, 1, size=rp[-1]))
    return cases


CASES = csr_cases()


# kernel -> (LL_SPMV_KERNEL, LL_PB_BLOCK, LL_PB_PHASE2): the CSR-stream kernel and the propagation-blocked kernels in
# their three phase-2 forms, each with the default geometry (a few blocks at these sizes) and with tiny blocks (many
# row and column blocks, ragged last blocks, empty segments)
KERNELS = {"csr_stream": ("csr", None, None),
           "pb_fixed": ("pb", None, "fixed"), "pb_fixed_small_blocks": ("pb", "37", "fixed"),
           "pb_fixed_ragged_blocks": ("pb", "53", "fixed"),
           "pb_ordered": ("pb", None, "ordered"), "pb_ordered_small_blocks": ("pb", "37", "ordered"),
           "pb_atomic": ("pb", None, "atomic"), "pb_atomic_small_blocks": ("pb", "37", "atomic"),
           # the 2-D tiled kernel (LL_TL_FORCE: these small matrices are not all eligible on their own): default row blocks,
           # tiny and ragged row blocks (many row blocks per column tile, empty tiles, one-quad tiles)
           "tiled": ("tiled", None, None), "tiled_small_blocks": ("tiled", "37", None),
           "tiled_ragged_blocks": ("tiled", "53", None),
           # ... and in the component-wise class (round 6: the waves add in turn in floating point, like pb_phase2<ORDERED>)
           "tiled_ordered": ("tiled", None, "ordered"), "tiled_ordered_ragged_blocks": ("tiled", "53", "ordered")}
KIND = {"csr": 0, "pb": 1, "tiled": 2}


@pytest.mark.parametrize("name", sorted(CASES))
@pytest.mark.parametrize("offset", [0.0, -2.5])
@pytest.mark.parametrize("kernel", sorted(KERNELS))
def test_spmv_matches_oracle(ctx, oracle, name, offset, kernel, llenv):
    csr = CASES[name]
    dtype = csr[2].dtype
    n = csr[0].shape[0] - 1
    x = rnd(n, dtype, 3)
    which, block, phase2 = KERNELS[kernel]
    llenv.setenv("LL_SPMV_KERNEL", which)
    if which == "tiled":
        llenv.setenv("LL_TL_FORCE", "1")
    if block:
        llenv.setenv("LL_PB_BLOCK", block)
    if phase2:
        llenv.setenv("LL_PB_PHASE2", phase2)
    op = L.CsrOperator(ctx, *csr)
    assert op.selected_spmv() == KIND[which]
    xd, yd = ctx.to_device(x), ctx.empty(n, dtype)
    alpha = L.spmv(op, xd, yd, offset=offset, want_dot=True)
    y = yd.get()
    y_ref = oracle.spmv(csr, x) + offset * x
    rp = csr[0]
    import scipy.sparse as sp

    absrow = sp.csr_matrix((np.abs(csr[2]), csr[1], rp), shape=(n, n)) @ np.abs(x) + abs(offset) * np.abs(x)
    # |y - y_ref| <= c * nnz_row * eps * sum_j |a_ij||x_j|   (different summation order, fma contraction); x ~ U[-1, 1]:
    # the norm-wise bound of the fixed-point form coincides with this one here (test_gpu_round3.py separates them)
    assert np.all(np.abs(y - y_ref) <= 8 * EPS * (np.diff(rp) + 2) * absrow + 1e-300)
    alpha_ref = np.vdot(x, y_ref).real
    assert abs(alpha - alpha_ref) <= 1e-13 * max(1.0, np.sum(np.abs(x) * np.abs(y_ref)))
    # without the fused dot: same y, bit for bit (CSR-stream folds in a fixed order, the PB kernels add fixed-point
    # integers or wave by wave in a fixed order); only the arrival-order A/B variant may differ by rounding
    L.spmv(op, xd, yd, offset=offset)
    if not kernel.startswith("pb_atomic"):
        assert np.array_equal(yd.get(), y)
    else:
        assert np.all(np.abs(yd.get() - y) <= 8 * EPS * (np.diff(rp) + 2) * absrow + 1e-300)
    op.close()


@pytest.mark.parametrize("name", ["randsym5000", "ragged", "ragged_z", "torus24", "banded20000"])
def test_pb_fixed_point_sums_do_not_depend_on_the_block_geometry(ctx, name, llenv):
    """Integer addition is associative: the fixed-point phase 2 gives the same bits for every block geometry (hence for
    every partition of the matrix); the wave-ordered floating-point form agrees with it to rounding."""
    csr = CASES[name]
    dtype = csr[2].dtype
    n = csr[0].shape[0] - 1
    x = rnd(n, dtype, 11)
    xd, yd = ctx.to_device(x), ctx.empty(n, dtype)
    llenv.setenv("LL_SPMV_KERNEL", "pb")
    fx = {}
    for block in (None, "37", "53", "1000"):
        llenv.setenv("LL_PB_PHASE2", "fixed")
        if block:
            llenv.setenv("LL_PB_BLOCK", block)
        op = L.CsrOperator(ctx, *csr)
        L.spmv(op, xd, yd, offset=0.5)
        fx[block] = yd.get()
        op.close()
        llenv.delenv("LL_PB_BLOCK")
    for block in ("37", "53", "1000"):
        assert np.array_equal(fx[None], fx[block]), block
    # ... and the 2-D tiled kernel adds the SAME integers (it stores the values pre-scaled by the row's exponent, an exact
    # operation, instead of looking the exponent up per entry): its result equals the PB kernel's bit for bit, whatever the tiling
    llenv.setenv("LL_SPMV_KERNEL", "tiled")
    llenv.setenv("LL_TL_FORCE", "1")
    for block in (None, "37", "1000"):
        if block:
            llenv.setenv("LL_PB_BLOCK", block)
        op = L.CsrOperator(ctx, *csr)
        assert op.selected_spmv() == 2
        L.spmv(op, xd, yd, offset=0.5)
        assert np.array_equal(fx[None], yd.get()), ("tiled", block)
        op.close()
        llenv.delenv("LL_PB_BLOCK")
    llenv.setenv("LL_SPMV_KERNEL", "pb")
    llenv.setenv("LL_PB_PHASE2", "ordered")
    op = L.CsrOperator(ctx, *csr)
    L.spmv(op, xd, yd, offset=0.5)
    yo = yd.get()
    L.spmv(op, xd, yd, offset=0.5)
    assert np.array_equal(yd.get(), yo)  # the wave-ordered form is reproducible launch to launch
    op.close()
    assert np.max(np.abs(fx[None] - yo)) <= 64 * EPS * np.max(np.abs(yo))


@pytest.mark.parametrize("name", ["randsym5000", "torus24", "ragged", "ragged_z"])
def test_spmv_with_64bit_row_offsets(ctx, oracle, name, llenv):
    """The int64 row_ptr variant of the CSR-stream kernel (used once nnz exceeds 2^31) on small matrices."""
    llenv.setenv("LL_FORCE_RP64", "1")
    csr = CASES[name]
    dtype = csr[2].dtype
    n = csr[0].shape[0] - 1
    x = rnd(n, dtype, 9)
    llenv.setenv("LL_SPMV_KERNEL", "csr")
    op = L.CsrOperator(ctx, *csr)
    xd, yd = ctx.to_device(x), ctx.empty(n, dtype)
    alpha = L.spmv(op, xd, yd, offset=0.25, want_dot=True)
    y_ref = oracle.spmv(csr, x) + 0.25 * x
    assert np.max(np.abs(yd.get() - y_ref)) <= 1e-13 * 40
    assert abs(alpha - np.vdot(x, y_ref).real) <= 1e-11 * n
    op.close()


@pytest.mark.parametrize("name", ["randsym5000", "ragged_z"])
def test_pb_image_build_with_64bit_row_offsets(ctx, oracle, name, llenv):
    """The device-side image build (histogram + scatter + row exponents) reads int64 row offsets once nnz exceeds 2^31;
    forced here on small matrices, for the default and the fixed-point phase 2."""
    llenv.setenv("LL_FORCE_RP64", "1")
    llenv.setenv("LL_SPMV_KERNEL", "pb")
    csr = CASES[name]
    dtype = csr[2].dtype
    n = csr[0].shape[0] - 1
    x = rnd(n, dtype, 5)
    y_ref = oracle.spmv(csr, x) - 1.5 * x
    xd, yd = ctx.to_device(x), ctx.empty(n, dtype)
    for phase2 in ("ordered", "fixed"):
        llenv.setenv("LL_PB_PHASE2", phase2)
        op = L.CsrOperator(ctx, *csr)
        assert op.selected_spmv() == L.capi.SPMV_PB
        L.spmv(op, xd, yd, offset=-1.5)
        assert np.max(np.abs(yd.get() - y_ref)) <= 1e-13 * 60
        op.close()


def test_fixed_point_phase2_reports_non_finite_input_as_nan(ctx, llenv):
    """Order-independent integer sums cannot carry Inf / NaN; rows that meet one are reported as NaN, every other row
    keeps its value (with a finite max |x| the scale of the clean rows is unaffected by a NaN elsewhere)."""
    llenv.setenv("LL_SPMV_KERNEL", "pb")
    llenv.setenv("LL_PB_PHASE2", "fixed")
    csr = CASES["laplace37"]
    n = csr[0].shape[0] - 1
    x = rnd(n, np.float64, 2)
    op = L.CsrOperator(ctx, *csr)
    xd, yd = ctx.to_device(x), ctx.empty(n)
    L.spmv(op, xd, yd)
    clean = yd.get()
    bad = x.copy()
    bad[100] = np.nan
    xd.set(bad)
    L.spmv(op, xd, yd)
    got = yd.get()
    rp, ci, _ = csr
    touched = np.array([np.any(ci[rp[i]:rp[i + 1]] == 100) for i in range(n)])
    assert np.all(np.isnan(got[touched])) and touched.sum() == 5
    assert np.max(np.abs(got[~touched] - clean[~touched])) <= 1e-13 * np.max(np.abs(clean))
    bad[100] = np.inf                       # max |x| = Inf: no usable scale for any row
    xd.set(bad)
    L.spmv(op, xd, yd)
    assert np.all(np.isnan(yd.get()))
    op.close()


def test_device_memory_helpers(ctx):
    """ll_malloc / ll_memset / ll_memcpy_{h2d,d2h} / ll_free round trip."""
    import ctypes as C

    a = ctx.empty(1000)
    a.set(np.arange(1000.0))
    L.capi.check(L.capi.lib().ll_memset(ctx.handle, C.c_void_p(a.ptr + 8 * 100), 0, 8 * 300))
    got = a.get()
    want = np.arange(1000.0)
    want[100:400] = 0.0
    assert np.array_equal(got, want)
    a.free()


def test_inner_product_convention(ctx):
    """T1:47-59: <(3, 1+3i), (3, 2+4i)> = 23 - 2i — conjugate-linear in the FIRST argument (LA:41,49)."""
    a = ctx.to_device(np.array([3.0, 1 + 3j]))
    b = ctx.to_device(np.array([3.0, 2 + 4j]))
    assert L.dot(ctx, a, b) == complex(23.0, -2.0)


@pytest.mark.parametrize("dtype", [np.float64, np.complex128])
@pytest.mark.parametrize("n", [1, 7, 2048, 2049, 100003, 1 << 20])
def test_blas1(ctx, dtype, n):
    a, b = rnd(n, dtype, 11), rnd(n, dtype, 12)
    ad, bd = ctx.to_device(a), ctx.to_device(b)
    tol = 4 * EPS * n * 2
    assert abs(L.dot(ctx, ad, bd) - np.vdot(a, b)) <= tol                       # a3  LA:29-51
    assert abs(L.nrm2(ctx, ad) - np.linalg.norm(a)) <= tol                       # a7  LA:56-60
    L.scal(ctx, -0.75, bd)                                                       # a8  LA:65-72
    assert np.array_equal(bd.get(), -0.75 * b)
    nrm = L.normalize(ctx, ad)                                                   # a8  LA:77-80
    assert abs(nrm - np.linalg.norm(a)) <= tol
    assert np.allclose(ad.get(), a * (1.0 / np.linalg.norm(a)), rtol=4 * EPS, atol=0)
    # a4 three-term update LL:251-257
    w, up, uc = rnd(n, dtype, 21), rnd(n, dtype, 22), rnd(n, dtype, 23)
    wd, upd, ucd = ctx.to_device(w), ctx.to_device(up), ctx.to_device(uc)
    L.three_term(ctx, wd, upd, ucd, 0.3, -1.7)
    assert np.allclose(wd.get(), w - 0.3 * up - (-1.7) * uc, rtol=0, atol=8 * EPS)
    wd.set(w)
    L.three_term(ctx, wd, None, ucd, 0.0, 0.9)                                   # k == 1: no beta term
    assert np.allclose(wd.get(), w - 0.9 * uc, rtol=0, atol=8 * EPS)


def _orthonormal_basis(n, nb, dtype, seed):
    rng = np.random.default_rng(seed)
    m = rng.uniform(-1, 1, (n, nb))
    if np.dtype(dtype) == np.complex128:
        m = m + 1j * rng.uniform(-1, 1, (n, nb))
    q, _ = np.linalg.qr(m)
    return np.ascontiguousarray(q.T)


# geometry: the Gram-Schmidt kernels have a streaming geometry (64 B per lane, 4 vectors per trip; vectors >= 4 MiB) and a
# small-vector one (16 B per lane, 16 vectors per trip); LL_BLAS_SMALL_BYTES forces either on every size
@pytest.mark.parametrize("geometry", ["0", str(1 << 40)], ids=["streaming", "small"])
@pytest.mark.parametrize("dtype", [np.float64, np.complex128])
@pytest.mark.parametrize("mode", [L.ORTH_CGS_DGKS, L.ORTH_CGS2, L.ORTH_MGS])
@pytest.mark.parametrize("n,nb", [(10, 5), (4099, 1), (100003, 37), (30011, 700), (4099, 1700)])  # 1700 > one launch
def test_orth_block_matches_mgs_oracle(ctx, oracle, dtype, mode, n, nb, geometry, llenv):
    """a5/a6/a7: block Gram-Schmidt vs the reference's sequential MGS (LA:132-144, test T1:61-91)."""
    llenv.setenv("LL_BLAS_SMALL_BYTES", geometry)
    basis = _orthonormal_basis(n, nb, dtype, 5)
    w = rnd(n, dtype, 31) + 3.0 * basis[0] - 2.0 * basis[nb - 1]
    ld = ((n + 255) // 256) * 256
    slab = np.zeros((nb, ld), dtype=dtype)
    slab[:, :n] = basis
    bd, wd = ctx.to_device(slab), ctx.to_device(w)
    nrm, h = L.orth_block(ctx, bd, nb, ld, wd, n, mode=mode, want_h=True)
    got = wd.get()
    want = oracle.schmidt_orth(basis.astype(np.complex128), w.astype(np.complex128))
    if np.dtype(dtype) == np.float64:
        want = want.real
    scale = np.linalg.norm(w)
    assert np.linalg.norm(got - want) <= 50 * EPS * scale * np.sqrt(nb)
    assert abs(nrm - np.linalg.norm(want)) <= 50 * EPS * scale * np.sqrt(nb)
    # residual overlaps (T1:86-90 uses 1e-15*n on vectors of norm ~30)
    ov = basis.conj() @ got
    assert np.max(np.abs(ov)) <= 20 * EPS * scale
    # the coefficients are the projections
    assert np.allclose(h, basis.conj() @ w, rtol=0, atol=50 * EPS * scale)


@pytest.mark.parametrize("dtype", [np.float64, np.complex128])
def test_orth_block_second_pass_triggers(ctx, dtype):
    """DGKS branch: w almost inside span(U) => the norm collapses and the predicated second pass must run;
    all three modes must agree on the (tiny) remainder direction to working accuracy."""
    n, nb = 50021, 12
    basis = _orthonormal_basis(n, nb + 1, dtype, 9)
    extra, basis = basis[nb], np.ascontiguousarray(basis[:nb])
    coeff = np.arange(1, nb + 1, dtype=np.float64)
    w = coeff @ basis + 1e-9 * extra
    ld = ((n + 255) // 256) * 256
    slab = np.zeros((nb, ld), dtype=dtype)
    slab[:, :n] = basis
    bd = ctx.to_device(slab)
    outs = []
    for mode in (L.ORTH_CGS_DGKS, L.ORTH_CGS2):
        wd = ctx.to_device(w)
        nrm = L.orth_block(ctx, bd, nb, ld, wd, n, mode=mode)
        got = wd.get()
        assert abs(nrm - 1e-9) <= 1e-6 * 1e-9 + 50 * EPS * np.linalg.norm(w)
        assert abs(np.linalg.norm(got) - nrm) <= 1e-3 * nrm
        # after "twice is enough" the remainder is orthogonal to U relative to ITS OWN norm
        assert np.max(np.abs(basis.conj() @ got)) <= 1e-6 * nrm
        outs.append(got)
    assert np.linalg.norm(outs[0] - outs[1]) <= 1e-6 * 1e-9


@pytest.mark.parametrize("dtype", [np.float64, np.complex128])
@pytest.mark.parametrize("n,m,nout", [(1000, 3, 1), (100003, 41, 5), (5000, 600, 2)])
def test_gemv_basis(ctx, dtype, n, m, nout):
    """a9/a10: out_r = sum_k c[r,k] u_k in one pass (LL:51-57, EX:166-170)."""
    rng = np.random.default_rng(3)
    ld = ((n + 255) // 256) * 256
    slab = np.zeros((m, ld), dtype=dtype)
    slab[:, :n] = rng.uniform(-1, 1, (m, n))
    coeff = rng.uniform(-1, 1, (nout, m)).astype(dtype)
    if np.dtype(dtype) == np.complex128:
        slab[:, :n] += 1j * rng.uniform(-1, 1, (m, n))
        coeff = coeff + 1j * rng.uniform(-1, 1, (nout, m))
    bd = ctx.to_device(slab)
    od = ctx.empty((nout, ld), dtype)
    L.gemv_basis(ctx, bd, m, ld, coeff, od, ld, n)
    got = od.get()[:, :n]
    want = coeff @ slab[:, :n]
    assert np.max(np.abs(got - want)) <= 8 * EPS * m * 2
