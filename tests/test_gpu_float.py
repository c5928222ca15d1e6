"""float / complex<float> storage (_s / _c entry points): the reference supports them (LL:143-149, test T1:163-193).
All reductions are carried in double on the device and the k-sized host math is double, so results are at least as
accurate as the reference's float arithmetic; parity is checked against the double oracle on the float-rounded inputs
with tolerances scaled to the float machine epsilon."""
import numpy as np
import pytest

import cases
import lambda_lanczos_amd as L
from lambda_lanczos_amd import generators as G
from util import overlap

pytestmark = pytest.mark.gpu
F32 = float(np.finfo(np.float32).eps)
REAL = {np.dtype(np.float32): np.float64, np.dtype(np.complex64): np.complex128}


def to_single(csr, dtype):
    return csr[0], csr[1], np.ascontiguousarray(csr[2]).astype(dtype)


def widen(csr):
    return csr[0], csr[1], csr[2].astype(REAL[csr[2].dtype])


@pytest.mark.parametrize("name,dtype", [("randsym", np.float32), ("laplace", np.float32), ("torus", np.complex64)])
@pytest.mark.parametrize("kind", [L.capi.SPMV_CSR_STREAM, L.capi.SPMV_PB])
def test_spmv_single_precision(ctx, oracle, name, dtype, kind, llenv):
    llenv.setenv("LL_PB_BLOCK", "257")
    llenv.setenv("LL_SPMV_KEEP_BOTH", "1")   # select_spmv below needs both images
    csr = to_single({"randsym": G.randsym_np(5000), "laplace": G.laplace2d_np(37), "torus": G.torus_np(24)}[name], dtype)
    n = csr[0].shape[0] - 1
    x = G.start_vector(n, 3, REAL[np.dtype(dtype)]).astype(dtype)
    op = L.CsrOperator(ctx, *csr)
    op.select_spmv(kind)
    xd, yd = ctx.to_device(x), ctx.empty(n, dtype)
    alpha = L.spmv(op, xd, yd, offset=-1.5, want_dot=True)
    y = yd.get()
    assert y.dtype == np.dtype(dtype)
    y_ref = oracle.spmv(widen(csr), x.astype(REAL[np.dtype(dtype)])) - 1.5 * x
    scale = np.max(np.abs(y_ref)) + 1.0
    assert np.max(np.abs(y - y_ref)) <= 40 * F32 * scale
    assert abs(alpha - np.vdot(x, y_ref).real) <= 40 * F32 * np.sum(np.abs(x) * np.abs(y_ref))
    op.close()


@pytest.mark.parametrize("dtype", [np.float32, np.complex64])
@pytest.mark.parametrize("n", [1, 9, 4097, 100003])
def test_blas1_single_precision(ctx, dtype, n):
    wide = REAL[np.dtype(dtype)]
    a, b = G.start_vector(n, 11, wide).astype(dtype), G.start_vector(n, 12, wide).astype(dtype)
    ad, bd = ctx.to_device(a), ctx.to_device(b)
    ref_dot = np.vdot(a.astype(wide), b.astype(wide))
    # reductions run in double on the device: far better than float accumulation
    assert abs(L.dot(ctx, ad, bd) - ref_dot) <= 1e-12 * n * 2
    assert abs(L.nrm2(ctx, ad) - np.linalg.norm(a.astype(wide))) <= 1e-12 * n
    nrm = L.normalize(ctx, ad)
    assert np.allclose(ad.get(), a / np.float32(nrm), rtol=4 * F32, atol=0)
    w, up, uc = (G.start_vector(n, s, wide).astype(dtype) for s in (21, 22, 23))
    wd, upd, ucd = ctx.to_device(w), ctx.to_device(up), ctx.to_device(uc)
    L.three_term(ctx, wd, upd, ucd, 0.3, -1.7)
    assert np.allclose(wd.get(), w - np.float32(0.3) * up + np.float32(1.7) * uc, rtol=0, atol=8 * F32)


# geometry: streaming (64 B per lane) and small-vector (16 B per lane, four waves per strip) Gram-Schmidt kernels, each
# forced on every size; the sizes leave partial 16-byte pieces at the end (n % 4 = 3, 2, 1)
@pytest.mark.parametrize("geometry", ["0", str(1 << 40)], ids=["streaming", "small"])
@pytest.mark.parametrize("n,nb", [(20011, 23), (66, 9), (7, 5), (513, 40)])
@pytest.mark.parametrize("dtype", [np.float32, np.complex64])
@pytest.mark.parametrize("mode", [L.ORTH_CGS_DGKS, L.ORTH_MGS])
def test_orth_and_gemv_single_precision(ctx, dtype, mode, n, nb, geometry, llenv):
    llenv.setenv("LL_BLAS_SMALL_BYTES", geometry)
    wide = REAL[np.dtype(dtype)]
    rng = np.random.default_rng(4)
    m = rng.uniform(-1, 1, (n, nb)) + (1j * rng.uniform(-1, 1, (n, nb)) if wide == np.complex128 else 0)
    q, _ = np.linalg.qr(m)
    basis = np.ascontiguousarray(q.T).astype(dtype)
    w = (G.start_vector(n, 31, wide) + 3.0 * q[:, 0]).astype(dtype)
    ld = ((n + 255) // 256) * 256
    slab = np.zeros((nb, ld), dtype=dtype)
    slab[:, :n] = basis
    bd, wd = ctx.to_device(slab), ctx.to_device(w)
    nrm, h = L.orth_block(ctx, bd, nb, ld, wd, n, mode=mode, want_h=True)
    got = wd.get().astype(wide)
    bw = basis.astype(wide)
    want = w.astype(wide) - (bw.conj() @ w.astype(wide)) @ bw
    scale = np.linalg.norm(w)
    assert np.linalg.norm(got - want) <= 20 * F32 * scale
    assert abs(nrm - np.linalg.norm(want)) <= 20 * F32 * scale
    assert np.max(np.abs(bw.conj() @ got)) <= 20 * F32 * scale
    # gemv over the same slab
    coeff = (rng.uniform(-1, 1, (3, nb)) + (1j * rng.uniform(-1, 1, (3, nb)) if wide == np.complex128 else 0)).astype(wide)
    od = ctx.empty((3, ld), dtype)
    L.gemv_basis(ctx, bd, nb, ld, coeff, od, ld, n)
    assert np.max(np.abs(od.get()[:, :n] - coeff @ bw)) <= 20 * F32 * nb


@pytest.mark.parametrize("dtype", [np.float32, np.complex64])
def test_engines_single_precision(ctx, oracle, dtype):
    wide = REAL[np.dtype(dtype)]
    # T1:163-193 SIMPLE_MATRIX_FLOAT: default eps = 1e3 * FLT_EPSILON, eigenvalue within |lambda| * eps
    op = L.CsrOperator(ctx, *to_single(G.dense_to_csr(cases.M3), dtype))
    eng = L.LambdaLanczos(op, 3, True, 1)
    assert abs(eng.eps - 1e3 * F32) <= 1e-12
    vals, vecs = eng.run()
    assert vecs.dtype == np.dtype(dtype)
    assert abs(vals[0] - 4.0) <= 4.0 * eng.eps
    assert np.max(np.abs(np.abs(vecs[0]) - 1 / np.sqrt(3))) <= 4.0 * eng.eps * 10
    op.close()
    # a larger problem against the double oracle on the float-rounded matrix
    csr = to_single(G.randsym_np(4096) if wide == np.float64 else G.torus_np(24), dtype)
    n = csr[0].shape[0] - 1
    init = G.start_vector(n, 1, wide).astype(dtype)
    find_max, offset = (True, 0.0) if wide == np.float64 else (False, -10.0)
    op = L.CsrOperator(ctx, *csr)
    eng = L.LambdaLanczos(op, n, find_max, 2)
    eng.eigenvalue_offset = offset
    eng.init_vector = lambda v, *_: np.copyto(v, init)
    vals, vecs = eng.run()
    ora = oracle.lanczos(widen(csr), init.astype(wide), find_max, num_eigs=2, offset=offset, eps=eng.eps)
    scale = max(1.0, np.max(np.abs(ora["eigenvalues"] + offset)))
    assert np.max(np.abs(vals - ora["eigenvalues"])) <= 20 * eng.eps * scale
    assert 1 - overlap(vecs[0].astype(wide), ora["eigenvectors"][0]) <= 1e-3
    # Exponentiator
    inp = G.start_vector(n, 2, wide).astype(dtype)
    a = -0.7j if wide == np.complex128 else -0.3
    ex = L.Exponentiator(op, n)
    assert abs(ex.eps - 1e2 * F32) <= 1e-12
    # the reference's stop test |1 - |<c_prev, c>|| < eps (EX:154) only fires for norm-preserving exponents; a real
    # exponent runs to max_iteration, so bound it like a user would
    ex.max_iteration = 40
    out, it = ex.run(a, inp)
    o_out, o_it, _ = oracle.expo(widen(csr), a, inp.astype(wide), eps=ex.eps, max_iteration=40)
    assert out.dtype == np.dtype(dtype) and abs(it - o_it) <= 2
    assert np.linalg.norm(out - o_out) <= 1e-3 * np.linalg.norm(o_out)
    t_out, terms = ex.taylor_run(a, inp)
    assert np.linalg.norm(t_out - o_out) <= 1e-3 * np.linalg.norm(o_out)
    op.close()
