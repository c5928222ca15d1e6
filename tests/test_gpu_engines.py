"""Whole-loop parity of the HIP path (through the C ABI) with the CPU oracle, the real reference (when
oracle/_ref/libref.so travelled) and the reference's own known answers (T1/T2), plus size-independent properties
at BASELINE sizes.

Stated fp64 tolerances:
  eigenvalue        |l_gpu - l_ref| <= 1e-10 * max(1, |l + offset|)
  eigenvector       1 - |<v_ref, v_gpu>| <= 1e-8
  alpha/beta trace  |d alpha_k|, |d beta_k| <= 1e-10 * ||A||_inf for every k
  iteration count   within +-2 of the oracle when run to convergence
  exponentiator     1 - overlap <= 10 * eps_engine ; |  ||out||/||in|| - 1 | <= 1e-12 for anti-Hermitian exponents
"""
import math

import numpy as np
import pytest

import cases
import lambda_lanczos_amd as L
from lambda_lanczos_amd import generators as G
from util import inf_norm, overlap, residual

pytestmark = pytest.mark.gpu
EPS = np.finfo(np.float64).eps


def fixed_init(vec):
    return lambda v, *_: v.__setitem__(slice(None), vec)


def gpu_engine(ctx, csr, find_max, k, **fields):
    op = L.CsrOperator(ctx, *csr)
    eng = L.LambdaLanczos(op, csr[0].shape[0] - 1, find_max, k)
    for key, val in fields.items():
        setattr(eng, key, val)
    return eng, op


# ------------------------------------------------------------------ the reference's known-answer tests (T1:128-536)
# geometry "streaming" (LL_BLAS_SMALL_BYTES=0) puts these small problems on the streaming kernels and, in the block
# Gram-Schmidt mode, on the one-sweep form with its locked-vector and breakdown paths
@pytest.mark.parametrize("geometry", ["default", "streaming"])
@pytest.mark.parametrize("name", sorted(cases.eigen_cases()))
@pytest.mark.parametrize("orth_mode", [L.ORTH_CGS_DGKS, L.ORTH_MGS])
def test_reference_known_answers(ctx, oracle, name, orth_mode, geometry, llenv):
    if geometry == "streaming":
        if orth_mode == L.ORTH_MGS:
            pytest.skip("the sequential mode has one geometry-independent code path per vector")
        llenv.setenv("LL_BLAS_SMALL_BYTES", "0")
    case = cases.eigen_cases()[name]
    csr = case["csr"]
    n = csr[0].shape[0] - 1
    dtype = csr[2].dtype
    init = G.start_vector(n, 1, dtype)
    eng, op = gpu_engine(ctx, csr, case["find_maximum"], case["num_eigs"], eigenvalue_offset=case["offset"],
                         init_vector=fixed_init(init), orth_mode=orth_mode)
    if case["eps"] is not None:
        eng.eps = case["eps"]
    vals, vecs = eng.run()
    eps_eng = eng.eps
    assert len(vals) == case["num_eigs"]
    for i, lam in enumerate(case["values"]):
        tol = case.get("abs_tol") or max(abs(lam) * eps_eng, 1e-8 if eps_eng > 1e-8 else 0.0)    # T1:156,478,532
        assert abs(vals[i] - lam) <= max(tol, 4 * EPS * abs(lam + case["offset"])), (name, i, vals[i], lam)
    if case["vectors"] is not None:
        for i, want in enumerate(case["vectors"]):
            got = vecs[i]
            phase = np.vdot(want, got)
            phase /= abs(phase)                                                                   # T1:150, 396-401
            tol = max(abs(case["values"][i]) * eps_eng * 10, 5e-8 if eps_eng > 1e-8 else 0.0)
            assert np.max(np.abs(got - phase * want)) <= max(tol, 1e-12), (name, i)
    # same problem, same start vector through the oracle: same answers, same iteration counts (+-2 per pass)
    ora = oracle.lanczos(csr, init, case["find_maximum"], num_eigs=case["num_eigs"], eps=case["eps"],
                         offset=case["offset"])
    assert len(ora["eigenvalues"]) == len(vals)
    assert np.max(np.abs(vals - ora["eigenvalues"])) <= 1e-10 * max(1.0, np.max(np.abs(vals + case["offset"])))
    if case["num_eigs"] == 1:
        assert len(eng.getIterationCounts()) == 1                                                # T1:160
        assert abs(eng.getIterationCounts()[0] - ora["iter_counts"][0]) <= 2
        assert 1 - overlap(vecs[0], ora["eigenvectors"][0]) <= 1e-8
    op.close()


def test_single_pair_overload_restores_num_eigs(ctx):
    """run(eigenvalue, eigenvector) computes one pair regardless of num_eigs and leaves it untouched (LL:394-407)."""
    csr = G.dense_to_csr(cases.M8)
    eng, op = gpu_engine(ctx, csr, False, 3, eps=1e-7)
    lam, vec = eng.run_single()
    assert eng.num_eigs == 3 and vec.shape == (8,)
    assert abs(lam - cases.M8_VALS[0]) <= 1e-6
    op.close()


def test_default_random_start_vector(ctx):
    """T1:195-229 (..._NOT_FIX_RANDOM_SEED): the default init_vector is random but the answer is not."""
    csr = G.dense_to_csr(cases.M3)
    out = []
    for _ in range(3):
        eng, op = gpu_engine(ctx, csr, True, 1, eigenvalue_offset=6.0)
        vals, vecs = eng.run()
        out.append(vecs[0])
        assert abs(vals[0] - 4.0) <= 4.0 * eng.eps
        assert overlap(vecs[0], np.ones(3)) >= 1 - 1e-12
        op.close()


def test_host_callback_operator_readme_sample(ctx):
    """BASELINE config 1: the README / sample1 3x3 dense lambda, unmodified user code through the host callback."""
    calls = []

    def mv_mul(inp, out):  # src/samples/sample1_simple.cpp:22-28 — accumulates into the zero-filled out
        assert np.all(out == 0)
        calls.append(1)
        for i in range(3):
            for j in range(3):
                out[i] += cases.M3[i][j] * inp[j]

    eng = L.LambdaLanczos(mv_mul, 3, True, 1, context=ctx)
    vals, vecs = eng.run()
    assert abs(vals[0] - 4.0) <= 1e-12 and overlap(vecs[0], np.ones(3)) >= 1 - 1e-12
    assert len(calls) >= eng.getIterationCounts()[0]

    def mv_overwrite(inp, out):  # sample4_use_Eigen_library.cpp:29 overwrites instead of accumulating
        out[:] = cases.M3 @ inp

    vals2, _ = L.LambdaLanczos(mv_overwrite, 3, True, 1, context=ctx).run()
    assert abs(vals2[0] - 4.0) <= 1e-12


# ------------------------------------------------------------------ traces and convergence vs oracle / reference
TRACE_CASES = {
    "laplace64_fixed40": dict(csr=lambda: G.laplace2d_np(64), find_max=False, offset=-8.0, max_iteration=40),
    "laplace64_converge": dict(csr=lambda: G.laplace2d_np(64), find_max=False, offset=-8.0, max_iteration=None),
    "randsym4096": dict(csr=lambda: G.randsym_np(4096), find_max=True, offset=0.0, max_iteration=None),
    "banded20000_fixed60": dict(csr=lambda: G.randsym_np(20000, band=512), find_max=True, offset=0.0, max_iteration=60),
    "torus16_hermitian": dict(csr=lambda: G.torus_np(16), find_max=False, offset=-10.0, max_iteration=None),
}


@pytest.mark.parametrize("name", sorted(TRACE_CASES))
@pytest.mark.parametrize("tridiag_mode", [L.TRIDIAG_QR, L.TRIDIAG_AUTO])
def test_traces_match_oracle(ctx, oracle, name, tridiag_mode):
    spec = TRACE_CASES[name]
    csr = spec["csr"]()
    n = csr[0].shape[0] - 1
    dtype = csr[2].dtype
    init = G.start_vector(n, 1, dtype)
    eng, op = gpu_engine(ctx, csr, spec["find_max"], 1, eigenvalue_offset=spec["offset"],
                         init_vector=fixed_init(init), tridiag_mode=tridiag_mode)
    if spec["max_iteration"]:
        eng.max_iteration = spec["max_iteration"]
    vals, vecs = eng.run()
    ora = oracle.lanczos(csr, init, spec["find_max"], offset=spec["offset"], max_iteration=spec["max_iteration"])
    it_gpu, it_ora = eng.getIterationCounts()[0], ora["iter_counts"][0]
    assert abs(it_gpu - it_ora) <= 2
    m = min(it_gpu, it_ora)
    anorm = inf_norm(csr) + abs(spec["offset"])
    assert np.max(np.abs(eng.last_alpha[:m] - ora["alpha"][:m])) <= 1e-10 * anorm
    assert np.max(np.abs(eng.last_beta[: m - 1] - ora["beta"][: m - 1])) <= 1e-10 * anorm
    assert abs(vals[0] - ora["eigenvalues"][0]) <= 1e-10 * max(1.0, abs(vals[0] + spec["offset"]))
    assert 1 - overlap(vecs[0], ora["eigenvectors"][0]) <= 1e-8
    if not spec["max_iteration"]:
        assert residual(csr, vals[0], vecs[0]) <= 1e-6 * anorm
    op.close()


def test_matches_real_reference(ctx, reference):
    """Same seeded problem through the REAL reference headers (oracle/_ref/libref.so, prebuilt)."""
    csr = G.randsym_np(3000)
    init = G.start_vector(3000)
    eng, op = gpu_engine(ctx, csr, True, 2, init_vector=fixed_init(init))
    vals, vecs = eng.run()
    ref = reference.lanczos(csr, init, True, num_eigs=2)
    assert np.max(np.abs(vals - ref["eigenvalues"])) <= 1e-10 * np.max(np.abs(vals))
    for i in range(2):
        assert 1 - overlap(vecs[i], ref["eigenvectors"][i]) <= 1e-8
    assert len(eng.getIterationCounts()) == len(ref["iter_counts"])
    op.close()


# ------------------------------------------------------------------ Exponentiator (T2:31-222)
@pytest.mark.parametrize("name", sorted(cases.expo_cases()))
def test_exponentiator_known_answers(ctx, oracle, name):
    case = cases.expo_cases()[name]
    csr = case["csr"]
    n = csr[0].shape[0] - 1
    op = L.CsrOperator(ctx, *csr)
    ex = L.Exponentiator(op, n)
    ex.full_orthogonalize = case["full"]
    out, itern = ex.run(case["a"], case["input"])
    assert abs(1 - overlap(case["exact"], out)) <= 10 * ex.eps + 4 * EPS                       # T2:66-72
    o_out, o_it, _ = oracle.expo(csr, case["a"], case["input"], full_orthogonalize=case["full"])
    assert abs(itern - o_it) <= 1
    assert np.max(np.abs(out - o_out)) <= 1e-10 * np.linalg.norm(o_out)
    t_out, terms = ex.taylor_run(case["a"], case["input"])                                     # T2:74-80
    assert abs(1 - overlap(case["exact"], t_out)) <= 10 * ex.eps + 4 * EPS
    ot_out, ot_terms, _ = oracle.expo(csr, case["a"], case["input"], taylor=True)
    assert terms == ot_terms
    op.close()


@pytest.mark.parametrize("dt", [0.1, 1.0, 5.0])
def test_exponentiator_torus_unitary(ctx, oracle, dt):
    """Config 5 in small: complex Hermitian torus, a = -i*dt: norm preserving, matches the oracle."""
    csr = G.torus_np(32)
    inp = G.start_vector(1024, 1, np.complex128)
    op = L.CsrOperator(ctx, *csr)
    ex = L.Exponentiator(op, 1024)
    out, itern = ex.run(-1j * dt, inp)
    o_out, o_it, _ = oracle.expo(csr, -1j * dt, inp)
    assert abs(itern - o_it) <= 1
    assert abs(np.linalg.norm(out) / np.linalg.norm(inp) - 1) <= 1e-12
    assert 1 - overlap(out, o_out) <= 10 * ex.eps
    assert np.max(np.abs(out - o_out)) <= 1e-10 * np.linalg.norm(inp)
    op.close()


# ------------------------------------------------------------------ BASELINE sizes: size-independent properties
def test_c2_laplacian_1M_properties(ctx):
    """Config 2 at full size (n = 1e6): fixed window; Lanczos relation ||A v - theta v|| = beta_m |s_m| and the
    Ritz value against the analytic spectrum bounds."""
    N = 1000
    csr = G.laplace2d(N)
    n = N * N
    init = G.start_vector_fast(n, 1)
    eng, op = gpu_engine(ctx, csr, False, 1, eigenvalue_offset=-8.0, init_vector=fixed_init(init), max_iteration=60)
    vals, vecs = eng.run()
    al, be = eng.last_alpha, eng.last_beta
    t = np.diag(al) + np.diag(be[:-1], 1) + np.diag(be[:-1], -1)
    w, s = np.linalg.eigh(t)
    theta = w[0] + 8.0
    assert abs(vals[0] - theta) <= 1e-10 * 8
    lam_min = G.laplace2d_lambda_min(N)
    assert lam_min - 1e-9 <= vals[0] <= 8.0                                  # Ritz values lie inside the spectrum
    assert abs(np.linalg.norm(vecs[0]) - 1) <= 1e-12
    res = residual(csr, vals[0], vecs[0])
    assert abs(res - be[-1] * abs(s[-1, 0])) <= 1e-9 * 8
    op.close()


def test_c3_random_10M_spmv_properties(ctx):
    """Config 3 at full size (n = 1e7, nnz = 1.5e8): SpMV linearity, symmetry <x,Ay> = <Ax,y> and A*1 = row sums."""
    n = 10_000_000
    csr = G.randsym(n)
    assert csr[0][-1] == 15 * n
    op = L.CsrOperator(ctx, *csr)
    x, y = G.start_vector_fast(n, 2), G.start_vector_fast(n, 3)
    xd, yd, sd = ctx.to_device(x), ctx.to_device(y), ctx.to_device(x + 2.0 * y)
    ax, ay, as_ = ctx.empty(n), ctx.empty(n), ctx.empty(n)
    L.spmv(op, xd, ax)
    L.spmv(op, yd, ay)
    L.spmv(op, sd, as_)
    axh, ayh = ax.get(), ay.get()
    assert np.max(np.abs(as_.get() - (axh + 2.0 * ayh))) <= 200 * EPS * 30
    assert abs(L.dot(ctx, xd, ay) - L.dot(ctx, ax, yd)) <= 1e-12 * n
    ones = ctx.to_device(np.ones(n))
    L.spmv(op, ones, as_)
    rowsum = np.add.reduceat(csr[2], csr[0][:-1])
    assert np.max(np.abs(as_.get() - rowsum)) <= 100 * EPS * 30
    op.close()


def test_tridiag_auto_reproduces_qr_decisions(ctx):
    """LL_TRIDIAG_AUTO (Sturm bisection per iteration, O(k)) hands the stop decision to the reference's QR arithmetic
    whenever a root's change comes within 4*eps of the threshold: iteration count and returned eigenvalue equal the
    reference-faithful LL_TRIDIAG_QR mode's exactly, on a run of several hundred iterations (5-point Laplacian
    200x200: the configuration SURVEY 8d quotes at 708 iterations)."""
    side = 200
    n = side * side
    csr = G.laplace2d_np(side)
    init = G.start_vector(n, 1)
    got = {}
    for mode in (L.TRIDIAG_QR, L.TRIDIAG_AUTO):
        eng, op = gpu_engine(ctx, csr, False, 1, eigenvalue_offset=-8.0, init_vector=fixed_init(init), tridiag_mode=mode)
        vals, vecs = eng.run()
        got[mode] = (eng.getIterationCounts(), float(vals[0]), vecs[0], eng.last_stats["seconds_host_tridiag"])
        op.close()
    assert got[L.TRIDIAG_QR][0] == got[L.TRIDIAG_AUTO][0] and got[L.TRIDIAG_QR][0][0] > 300
    assert got[L.TRIDIAG_QR][1] == got[L.TRIDIAG_AUTO][1]                 # the QR values are returned in both modes
    assert 1 - overlap(got[L.TRIDIAG_QR][2], got[L.TRIDIAG_AUTO][2]) <= 1e-10
    lam = G.laplace2d_lambda_min(side)
    assert abs(got[L.TRIDIAG_AUTO][1] - lam) <= 1e-10 * 8
    assert got[L.TRIDIAG_AUTO][3] < got[L.TRIDIAG_QR][3]                  # and the host step is cheaper


# ------------------------------------------------------------------ the host-decided second Gram-Schmidt pass
@pytest.mark.parametrize("geometry", ["default", "streaming"])
def test_exhausted_krylov_space_runs_like_the_oracle(ctx, oracle, geometry, llenv):
    """An operator with 5 distinct eigenvalues exhausts its Krylov space after 5 iterations: from then on w is rounding
    noise (beta ~ 1e-15, just above the breakdown threshold) and block Gram-Schmidt with the host-decided DGKS test
    must carry the run on exactly like the oracle's sequential MGS (same iteration count, same eigenpair)."""
    if geometry == "streaming":   # the one-sweep (lagged) Gram-Schmidt form: the noise iterations sit on its DGKS repair path
        llenv.setenv("LL_BLAS_SMALL_BYTES", "0")
    rng = np.random.default_rng(4)
    n = 300
    q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    lam = np.repeat([1.0, 2.0, 3.5, 5.0, 9.0], n // 5)
    a = (q * lam) @ q.T
    a = (a + a.T) / 2
    init = G.start_vector(n, 1)
    op = L.DenseOperator(ctx, a)
    eng = L.LambdaLanczos(op, n, True, 1)
    eng.init_vector = fixed_init(init)
    vals, vecs = eng.run()
    ora = oracle.lanczos(G.dense_to_csr(a), init, True)
    assert abs(vals[0] - 9.0) <= 1e-10 and abs(ora["eigenvalues"][0] - 9.0) <= 1e-10
    assert abs(eng.getIterationCounts()[0] - ora["iter_counts"][0]) <= 1
    assert eng.last_stats["second_passes"] >= 0   # reported; whether the noise iterations trigger it is data dependent
    assert np.linalg.norm(a @ vecs[0] - vals[0] * vecs[0]) <= 1e-9
    op.close()


@pytest.mark.parametrize("geometry", ["default", "streaming"])
@pytest.mark.parametrize("what", ["lanczos_two_roots", "expo_full_orth", "lanczos_complex"])
def test_forced_second_pass_every_iteration(ctx, oracle, what, geometry, llenv):
    """LL_DGKS_THRESHOLD > 1 makes the host decide for a second pass in EVERY iteration: the pipeline is drained,
    Gram-Schmidt is repeated on the normalised vector, beta is rescaled and the speculative next iteration is enqueued
    again.  A second pass on an already orthogonal vector changes nothing but rounding, so traces, iteration counts and
    results must still equal the oracle's."""
    llenv.setenv("LL_DGKS_THRESHOLD", "2.0")
    if geometry == "streaming":   # streaming kernels => the one-sweep (lagged) Gram-Schmidt form: every speculative
        llenv.setenv("LL_BLAS_SMALL_BYTES", "0")   # sweep is thrown away and the iteration redone from the repaired vector
    if what == "expo_full_orth":
        csr = G.torus_np(20)
        inp = G.start_vector(400, 1, np.complex128)
        op = L.CsrOperator(ctx, *csr)
        ex = L.Exponentiator(op, 400)
        ex.full_orthogonalize = True
        out, it = ex.run(-1j, inp)
        o_ref, it_ref, _ = oracle.expo(csr, -1j, inp, full_orthogonalize=True)
        # (a speculative iteration collected before the stop verdict arrived gets its second pass too, then is dropped)
        assert it == it_ref and it <= ex.last_stats["second_passes"] <= it + 2
        assert np.max(np.abs(out - o_ref)) <= 1e-11 * np.linalg.norm(inp)
        op.close()
        return
    csr = G.torus_np(24) if what == "lanczos_complex" else G.randsym_np(5000)
    n = csr[0].shape[0] - 1
    k = 1 if what == "lanczos_complex" else 2
    init = G.start_vector(n, 1, csr[2].dtype)
    eng, op = gpu_engine(ctx, csr, True, k, init_vector=fixed_init(init), max_iteration=60)
    vals, vecs = eng.run()
    ora = oracle.lanczos(csr, init, True, num_eigs=k, max_iteration=60)
    assert eng.getIterationCounts() == ora["iter_counts"]
    assert eng.last_stats["second_passes"] == sum(ora["iter_counts"])
    m = len(ora["alpha"])
    assert np.max(np.abs(eng.last_alpha[:m] - ora["alpha"])) <= 1e-10 * inf_norm(csr)
    assert np.max(np.abs(eng.last_beta[:m - 1] - ora["beta"][:m - 1])) <= 1e-10 * inf_norm(csr)
    assert np.max(np.abs(vals - ora["eigenvalues"])) <= 1e-10 * np.max(np.abs(vals))
    for i in range(k):
        assert 1 - overlap(vecs[i], ora["eigenvectors"][i]) <= 1e-8
    op.close()


# ------------------------------------------------------------------ run_iteration called directly (LL:216-322)
@pytest.mark.parametrize("name", ["m8_three_roots", "m8_lowest_locked", "randsym600_top2_locked", "torus12_lowest_locked"])
@pytest.mark.parametrize("orth_mode", [L.ORTH_CGS_DGKS, L.ORTH_MGS])
def test_run_iteration_matches_reference_fixture(ctx, oracle, name, orth_mode):
    """ll_lanczos_run_iteration_*: one pass, nroot pairs, the caller's orthogonalizeTo list — against the real
    reference's output (tests/golden/run_iteration.json) and the oracle."""
    from util import list2c, load_golden

    fx = load_golden("run_iteration.json")[name]
    csr, init = cases.run_iteration_problem(name)
    orth = None if fx["orth"] is None else np.array([list2c(v) for v in fx["orth"]])
    eng, op = gpu_engine(ctx, csr, fx["find_maximum"], 1, eigenvalue_offset=fx["offset"], init_vector=fixed_init(init),
                         orth_mode=orth_mode)
    vals, vecs, itern = eng.run_iteration(fx["nroot"], orth)
    want = np.array(fx["eigenvalues"])
    assert len(vals) == len(want) and abs(itern - fx["itern"]) <= 2
    assert eng.getIterationCounts() == [itern]
    assert np.max(np.abs(vals - want)) <= 1e-10 * max(1.0, np.max(np.abs(want + fx["offset"])))
    for got, ref_v in zip(vecs, fx["eigenvectors"]):
        assert 1 - overlap(got, list2c(ref_v)) <= 1e-8
    if orth is not None:
        assert np.max(np.abs(orth.conj() @ vecs.T)) <= 1e-8
    ora = oracle.run_iteration(csr, init, fx["find_maximum"], fx["nroot"], orth=orth, offset=fx["offset"])
    assert abs(itern - ora["itern"]) <= 2 and np.max(np.abs(vals - ora["eigenvalues"])) <= 1e-10 * max(1.0, np.max(np.abs(want)))
    op.close()


# ------------------------------------------------------------------ sharded code path on one GPU
def test_sharded_path_with_single_rank_communicator(oracle, llenv):
    """A 1-rank RCCL communicator drives the whole multi-GPU code path (dlopen of librccl, ncclCommInitRank, the
    all-gather of x before every SpMV and the all-reduces of alpha / Gram-Schmidt coefficients / norms on the
    library stream) on the single GPU of the test box: results must equal the communicator-free run."""
    llenv.setenv("LL_SPMV_KEEP_BOTH", "1")   # select_spmv below needs both images
    csr = G.randsym_np(30011)
    n = 30011
    init = G.start_vector(n)
    ctx2 = L.Context(0)
    ctx2.init_comm(L.Context.unique_id(), 0, 1)
    assert ctx2.partition(n) == (0, n)
    import ctypes as C

    r, w = C.c_int(-1), C.c_int(-1)
    L.capi.check(L.capi.lib().ll_comm_rank(ctx2.handle, C.byref(r), C.byref(w)))
    assert (r.value, w.value) == (0, 1)
    out = {}
    for label, c in (("comm", ctx2), ("plain", L.Context(0))):
        for kind in (L.capi.SPMV_CSR_STREAM, L.capi.SPMV_PB):
            op = L.CsrOperator(c, *csr)
            op.select_spmv(kind)
            eng = L.LambdaLanczos(op, n, True, 2)
            eng.init_vector = fixed_init(init)
            vals, vecs = eng.run()
            out[(label, kind)] = (vals, vecs, eng.getIterationCounts())
            op.close()
    ora = oracle.lanczos(csr, init, True, num_eigs=2)
    for key, (vals, vecs, counts) in out.items():
        assert np.max(np.abs(vals - ora["eigenvalues"])) <= 1e-10 * np.max(np.abs(vals)), key
        assert len(counts) == len(ora["iter_counts"]), key
        for i in range(2):
            assert 1 - overlap(vecs[i], ora["eigenvectors"][i]) <= 1e-8, key
    # complex + exponentiator through the communicator as well
    tcsr = G.torus_np(24)
    inp = G.start_vector(576, 1, np.complex128)
    top = L.CsrOperator(ctx2, *tcsr)
    o1, it1 = L.Exponentiator(top, 576).run(-1j, inp)
    o2, it2, _ = oracle.expo(tcsr, -1j, inp)
    assert abs(it1 - it2) <= 1 and np.max(np.abs(o1 - o2)) <= 1e-10 * np.linalg.norm(inp)
    top.close()
    ctx2.close()


def test_lattice_halo_exchange_with_single_rank_communicator(oracle):
    """The lattice operator's exchange step through RCCL itself: with a 1-rank communicator on a ring (periodic
    slowest dimension) both neighbours are the rank itself, so the grouped ncclSend/ncclRecv pair of
    comm_halo_exchange runs for real on the test box's single GPU; results must equal the communicator-free path."""
    dims = [12, 9, 7]
    n = int(np.prod(dims))
    kw = dict(diag=0.5, hop=[0.5 + 1j, -1.0, 0.25j], periodic=[True, False, True], dtype=np.complex128)
    csr = G.lattice_csr(dims, **kw)
    x = G.start_vector(n, 2, np.complex128)
    y_ref = oracle.spmv(csr, x)
    ctx2 = L.Context(0)
    ctx2.init_comm(L.Context.unique_id(), 0, 1)
    op = L.StencilOperator(ctx2, dims, **kw)
    xd, yd = ctx2.to_device(x), ctx2.empty(n, np.complex128)
    dot = L.spmv(op, xd, yd, want_dot=True)
    assert np.max(np.abs(yd.get() - y_ref)) <= 1e-13 * 10
    assert abs(dot - np.vdot(x, y_ref).real) <= 1e-11 * n
    out, it = L.Exponentiator(op, n).run(-0.5j, x)
    o2, it2, _ = oracle.expo(csr, -0.5j, x)
    assert abs(it - it2) <= 1 and np.max(np.abs(out - o2)) <= 1e-10 * np.linalg.norm(x)
    op.close()
    ctx2.close()


# ------------------------------------------------------------------ other operator forms of the mv_mul plugin
def test_device_array_csr_and_device_callback_operators(ctx, oracle, llenv):
    """ll_op_create_csr_dev_d (matrix already in HBM) and ll_op_create_device_d (a callback that enqueues
    out += A*in on the library stream for device pointers, `out` zero-filled) give the same run as the host-array CSR."""
    import ctypes as C

    from lambda_lanczos_amd import _capi as capi
    from lambda_lanczos_amd.engine import _Operator

    llenv.setenv("LL_SPMV_KERNEL", "pb")
    csr = G.randsym_np(20011)
    n = 20011
    init = G.start_vector(n)
    rp_d, ci_d, va_d = ctx.to_device(csr[0]), ctx.to_device(csr[1]), ctx.to_device(csr[2])

    class DevCsr(_Operator):
        pass

    dev = DevCsr()
    dev.ctx, dev.dtype, dev.n, dev.n_local, dev.row_begin, dev.nnz = ctx, np.dtype(np.float64), n, n, 0, int(csr[0][-1])
    h = C.c_void_p()
    capi.check(capi.lib().ll_op_create_csr_dev_d(ctx.handle, n, n, 0, rp_d.ptr, ci_d.ptr, va_d.ptr, C.byref(h)))
    dev.handle = h
    # device arrays get the propagation-blocked image too (built from a one-off copy back) and the row-sum bound
    capi.check(capi.lib().ll_op_select_spmv(h, capi.SPMV_PB))
    nrm = C.c_double()
    capi.check(capi.lib().ll_op_inf_norm(h, C.byref(nrm)))
    assert abs(nrm.value - inf_norm(csr)) <= 1e-12 * nrm.value

    calls = []

    def dev_mv(in_p, out_p, nn, stream, _user):  # user code working on device pointers: here another library call
        calls.append((nn, stream))
        tmp = ctx.empty(n)
        rc = capi.lib().ll_spmv_d(ctx.handle, dev.handle, in_p, tmp.ptr, 0.0, None)
        if rc:
            return rc
        out = L.DeviceArray.__new__(L.DeviceArray)
        # out += tmp  via three_term:  out = out - 0*prev - (-1)*tmp
        rc = capi.lib().ll_three_term_d(ctx.handle, nn, out_p, None, tmp.ptr, 0.0, -1.0)
        ctx.synchronize()
        tmp.free()
        return rc

    cb = capi.DEV_MV_FN(dev_mv)
    cbo = DevCsr()
    cbo.ctx, cbo.dtype, cbo.n, cbo.n_local, cbo.row_begin, cbo.nnz = ctx, np.dtype(np.float64), n, n, 0, 0
    h2 = C.c_void_p()
    capi.check(capi.lib().ll_op_create_device_d(ctx.handle, n, cb, None, C.byref(h2)))
    cbo.handle = h2

    ora = oracle.lanczos(csr, init, True, max_iteration=25)
    for op in (dev, cbo):
        eng = L.LambdaLanczos(op, n, True, 1)
        eng.max_iteration = 25
        eng.init_vector = fixed_init(init)
        vals, vecs = eng.run()
        assert abs(vals[0] - ora["eigenvalues"][0]) <= 1e-10 * abs(vals[0])
        assert np.max(np.abs(eng.last_alpha - ora["alpha"])) <= 1e-10 * 30
        assert 1 - overlap(vecs[0], ora["eigenvectors"][0]) <= 1e-8
    assert len(calls) >= 25 and calls[0][0] == n and calls[0][1] == ctx.stream()
    dev.close()
    cbo.close()


def test_context_on_a_torch_stream_with_torch_memory():
    """ll_ctx_create_on_stream: the library enqueues on a stream the application owns (a torch.cuda.Stream) and works on
    device memory it did not allocate (torch tensors) — torch as plumbing for memory and streams
    (tests/torch_stream_worker.py).  Runs in a child process: this pytest process also creates RCCL communicators, and a
    process that initialises torch.cuda AND loads ROCm's RCCL ends up with two copies of librocm_smi64 (torch's wheel
    bundles one) whose static destructors abort at interpreter exit — nothing to do with the library under test."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "torch_stream_worker.py")], capture_output=True, text=True,
                       timeout=300, cwd=root)
    assert r.returncode == 0 and "torch stream ok" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


def test_no_device_memory_growth_over_many_runs():
    """Operators, engines and contexts give their device memory back: create/run/destroy cycles (CSR with both SpMV
    images, lattice, dense; eigen solver with restart passes, Exponentiator) leave the free device memory where it was
    (40 cycles here; 150 were run once by hand)."""
    import ctypes as C

    hip = C.CDLL("libamdhip64.so.7")   # the instance the library already loaded (same SONAME)

    def free_bytes():
        free, total = C.c_size_t(), C.c_size_t()
        assert hip.hipDeviceSynchronize() == 0 and hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
        return free.value

    csr = G.randsym_np(20000)
    init = G.start_vector(20000, 1)
    tcsr = G.torus_np(40)
    tin = G.start_vector(1600, 1, np.complex128)
    dense = np.diag(np.arange(1.0, 301.0)) + 0.01

    def cycle():
        c = L.Context(0)
        op = L.CsrOperator(c, *csr)
        eng = L.LambdaLanczos(op, 20000, True, 3)
        eng.max_iteration = 30
        eng.init_vector = fixed_init(init)
        eng.run()
        op.close()
        top = L.CsrOperator(c, *tcsr)
        L.Exponentiator(top, 1600).run(-0.5j, tin)
        top.close()
        st = L.StencilOperator(c, [40, 40], diag=4.0)
        L.LambdaLanczos(st, 1600, False, 1).run()
        st.close()
        dn = L.DenseOperator(c, dense)
        L.LambdaLanczos(dn, 300, True, 1).run()
        dn.close()
        c.close()

    for _ in range(5):   # warm up allocator pools, code objects, RCCL-free paths
        cycle()
    before = free_bytes()
    for _ in range(40):
        cycle()
    after = free_bytes()
    assert before - after <= 64 << 20, (before, after)


def test_error_reporting(ctx):
    """Bad arguments come back as status codes with a message (the reference only asserts, LA:31, EX:88)."""
    csr = G.randsym_np(500)
    op = L.CsrOperator(ctx, *csr)
    eng = L.LambdaLanczos(op, 499, True, 1)          # wrong matrix_size
    with pytest.raises(L.LanczosHipError) as e:
        eng.run()
    assert e.value.code == L.capi.LL_ERR_INVALID and "matrix_size" in str(e.value)
    with pytest.raises(L.LanczosHipError):
        L.CsrOperator(ctx, csr[0], np.where(csr[1] == 3, 700, csr[1]), csr[2])   # column index out of range
    zop = L.CsrOperator(ctx, *G.torus_np(8))
    with pytest.raises(L.LanczosHipError):            # complex operator through the real entry point
        L.LambdaLanczos(zop, 64, True, 1, dtype=np.float64).run()

    def boom(a, b):
        raise RuntimeError("user code failed")

    with pytest.raises(L.LanczosHipError) as e2:
        L.LambdaLanczos(boom, 5, True, 1, context=ctx).run()
    assert e2.value.code == L.capi.LL_ERR_CALLBACK
    op.close()
    zop.close()


# ------------------------------------------------------------------ BASELINE config 5 at full size
@pytest.mark.parametrize("dt,expect_iters", [(0.1, 7), (1.0, 15), (5.0, 38)])
def test_c5_exponentiator_1M_properties(ctx, dt, expect_iters):
    """Complex Hermitian torus n = 1e6, a = -i*dt: unitary evolution preserves the norm to 1e-12; the iteration counts
    are the ones the real reference needed for this input (SURVEY 3.2 probe: 7 / 15 / 38); exp(-iH dt) exp(+iH dt) = 1."""
    N = 1000
    csr = G.torus(N)
    n = N * N
    inp = G.start_vector_fast(n, 1, np.complex128)
    op = L.CsrOperator(ctx, *csr)
    ex = L.Exponentiator(op, n)
    out, itern = ex.run(-1j * dt, inp)
    assert abs(itern - expect_iters) <= 1
    assert abs(np.linalg.norm(out) / np.linalg.norm(inp) - 1) <= 1e-12
    back, _ = ex.run(+1j * dt, out)
    # the engine stops on 1 - |<c_prev, c>| < eps = 2.2e-14 (EX:154), i.e. a coefficient error of ~sqrt(2 eps) ~ 2e-7
    assert np.linalg.norm(back - inp) <= 1e-6 * np.linalg.norm(inp)
    op.close()


def test_coo_operator_sample2_and_inf_norm_offset(ctx):
    """src/samples/sample2_sparse.cpp: {r, c, value} list of the 3x3 matrix with eigenvalues {1, 1, -2}, smallest wanted.
    And the offset helper: with eigenvalue_offset = -inf_norm the "smallest" problem converges."""
    rows, cols, vals = [0, 0, 1, 1, 2, 2], [1, 2, 0, 2, 0, 1], [1.0, 1.0, 1.0, -1.0, 1.0, -1.0]
    op = L.CsrOperator.from_coo(ctx, 3, rows, cols, vals)
    assert op.inf_norm() == 2.0
    eng = L.LambdaLanczos(op, 3, False, 1)
    vals_, vecs_ = eng.run()
    assert abs(vals_[0] + 2.0) <= 1e-12
    assert overlap(vecs_[0], np.array([1.0, -1.0, -1.0])) >= 1 - 1e-12
    op.close()
    csr = G.laplace2d_np(30)
    lap = L.CsrOperator(ctx, *csr)
    assert lap.inf_norm() == 8.0
    e2 = L.LambdaLanczos(lap, 900, False, 1)
    e2.eigenvalue_offset = -lap.inf_norm()
    v2, _ = e2.run()
    assert abs(v2[0] - G.laplace2d_lambda_min(30)) <= 1e-10 * 8
    assert e2.getIterationCounts()[0] < 900          # converged, did not run to max_iteration
    lap.close()


@pytest.mark.parametrize("geometry", ["default", "streaming"])
def test_profiled_runs_longer_than_the_timing_event_ring(geometry, llenv):
    """ll_ctx_set_profiling: the per-phase device times come from a ring of 128 event triples that the context keeps between
    runs (a triple is read back when its slot comes round again).  A run with several hundred iterations — more triples than
    the ring holds, in the pair form two per sweep — must report device times that add up (below the wall time of the call,
    above the bulk of it for a device-bound problem), the same figures for a second call on the same context, and the same
    Lanczos results as an unprofiled run."""
    if geometry == "streaming":
        llenv.setenv("LL_BLAS_SMALL_BYTES", "0")
    c = L.Context(0)
    m = 300
    csr = G.laplace2d_np(m)
    init = G.start_vector(m * m, 2)
    op = L.CsrOperator(c, *csr)

    def run():
        eng = L.LambdaLanczos(op, m * m, True, 1)
        eng.init_vector = fixed_init(init)
        vals, _ = eng.run()
        return vals[0], eng.getIterationCounts()[0], dict(eng.last_stats)

    plain = run()
    c.set_profiling(True)
    a, b = run(), run()
    c.set_profiling(False)
    assert plain[0] == a[0] == b[0] and plain[1] == a[1] == b[1] and a[1] > 3 * 128
    for st in (a[2], b[2]):
        dev = st["seconds_spmv"] + st["seconds_orth"]
        assert 0.0 < dev <= st["seconds_total"] * 1.02, st
        assert st["seconds_spmv"] > 0.0 and st["seconds_orth"] > 0.0
    da, db = a[2]["seconds_spmv"] + a[2]["seconds_orth"], b[2]["seconds_spmv"] + b[2]["seconds_orth"]
    assert da <= 3 * db and db <= 3 * da, (a[2], b[2])    # (loose: a host hiccup inside one triple lands in its device interval)
    op.close()
    c.close()
