"""Round-6 GPU tests (through the C ABI): per-context tuning instead of environment hooks, the transport query, the measured
streaming ceilings, the tiled kernel asked for by name where it cannot exist, the pair form's max_iteration bound."""
import os

import numpy as np
import pytest

import lambda_lanczos_amd as L
from lambda_lanczos_amd import _capi as capi
from lambda_lanczos_amd import generators as G

pytestmark = pytest.mark.gpu


def fixed_init(vec):
    return lambda v, *_: v.__setitem__(slice(None), vec)


# ------------------------------------------------------------------ ll_ctx_set_tuning
def test_set_tuning_rejects_unknown_keys_and_the_hooks_are_not_environment_switches(monkeypatch):
    """The test hooks are per-context settings: an unknown key is LL_ERR_INVALID, and a variable under an old hook name in the
    process environment does nothing — here LL_TL_FORCE, which used to make the tiled image buildable for a matrix that is not
    eligible (random columns): with the variable alone, asking for the tiled kernel by name is the documented error; with the
    context's own setting it is built, and gives the bits of the PB kernel."""
    from util import sync_hooks

    c2 = L.Context(0)
    try:
        with pytest.raises(capi.LanczosHipError) as e:
            c2.set_tuning("no_such_key", "1")
        assert e.value.code == capi.LL_ERR_INVALID and "unknown key" in str(e.value)
        n = 30000
        csr = G.randsym_np(n)
        monkeypatch.setenv("LL_TL_FORCE", "1")
        c2.reload_env()                      # the library reads its environment again: LL_TL_FORCE is not part of it
        c2.set_tuning("tl_force", None)      # (the harness had synced its hook names into the new context: undo)
        with pytest.raises(capi.LanczosHipError) as e:
            L.CsrOperator(c2, *csr, kernel=capi.SPMV_TILED)
        assert "not eligible" in str(e.value)
        sync_hooks(c2)                       # what the harness does with its hook names: ll_ctx_set_tuning(ctx, "tl_force", "1")
        op_t = L.CsrOperator(c2, *csr, kernel=capi.SPMV_TILED)
        op_p = L.CsrOperator(c2, *csr, kernel=capi.SPMV_PB)
        assert op_t.selected_spmv() == capi.SPMV_TILED and op_p.selected_spmv() == capi.SPMV_PB
        x = G.start_vector(n, 5)
        xd, ya, yb = c2.to_device(x / np.linalg.norm(x)), c2.empty(n), c2.empty(n)
        L.spmv(op_t, xd, ya)
        L.spmv(op_p, xd, yb)
        assert np.array_equal(ya.get(), yb.get())
        op_t.close()
        op_p.close()
    finally:
        c2.close()


def test_transport_query_without_a_communicator(ctx):
    assert ctx.transport() == "none"
    assert ctx.ranks_seen() == 1


def test_bandwidth_probe_reports_plausible_ceilings(ctx):
    rd, cp = ctx.bandwidth_probe(1 << 30)
    # MI355X: 8 TB/s spec; measured 6.3 TB/s read-only, ~5 TB/s copy (profiles/r01_bw_probe.txt)
    assert 2000.0 < rd < 8000.0 and 2000.0 < cp < 8000.0, (rd, cp)


# ------------------------------------------------------------------ tiled kernel by name where no tiled image can exist
def test_tiled_kernel_by_name_on_a_matrix_without_entries_is_an_error(ctx):
    """`kernel = LL_SPMV_TILED` used to select the kernel without an image when the matrix has no entries — y was
    never written.  Asking by name is an error, not a silent fallback."""
    n = 4096
    rp = np.zeros(n + 1, dtype=np.int64)
    with pytest.raises(capi.LanczosHipError) as e:
        L.CsrOperator(ctx, rp, np.zeros(0, dtype=np.int32), np.zeros(0), kernel=capi.SPMV_TILED)
    assert "tiled" in str(e.value)
    # the automatic choice still works on it (A = 0: y = offset x)
    op = L.CsrOperator(ctx, rp, np.zeros(0, dtype=np.int32), np.zeros(0))
    x = G.start_vector(n, 3)
    xd, yd = ctx.to_device(x), ctx.empty(n)
    L.spmv(op, xd, yd, offset=0.5)
    assert np.array_equal(yd.get(), 0.5 * x)
    op.close()


# ------------------------------------------------------------------ the pair form never runs past max_iteration
@pytest.mark.parametrize("window", [9, 10, 11])
def test_pair_form_honours_an_odd_max_iteration(ctx, oracle, window):
    """with an odd number of iterations left the pair form enqueued iteration max_iteration + 1 (one operator
    application more than asked for; Inf / NaN in the records when max_iteration == n).  The last odd iteration runs in the
    one-sweep form; alpha / beta and the Ritz pair are the oracle's."""
    n = 200_000
    csr = G.randsym(n)
    init = G.start_vector_fast(n, 1)
    op = L.CsrOperator(ctx, *csr)
    eng = L.LambdaLanczos(op, n, True, 1)
    eng.max_iteration = window
    eng.init_vector = fixed_init(init)
    vals, vecs = eng.run()
    st = eng.last_stats
    assert eng.getIterationCounts() == [window]
    assert st["pair_iterations"] % 2 == 0 and st["pair_iterations"] <= window - 2
    assert st["pair_iterations"] >= window - 3          # iterations 1, 2 single, then pairs, an odd last one single
    ora = oracle.lanczos(csr, init, True, max_iteration=window)
    assert np.max(np.abs(eng.last_alpha - ora["alpha"][:window])) <= 1e-10 * 30
    assert np.max(np.abs(eng.last_beta[: window - 1] - ora["beta"][: window - 1])) <= 1e-10 * 30
    assert abs(vals[0] - ora["eigenvalues"][0]) <= 1e-10 * 30
    assert np.all(np.isfinite(vecs[0]))
    op.close()


# ------------------------------------------------------------------ the pair form where it did not run
@pytest.mark.parametrize("side,dt", [(300, 5.0), (190, 5.0), (300, 1.0)])
def test_exponentiator_with_full_orthogonalize_takes_two_iterations_per_sweep(ctx, oracle, side, dt):
    """Exponentiator<T>::run with full_orthogonalize (EX:63,120-122) on config 5's matrix in small — complex torus 300 x 300 (1.44 MB
    vectors: streaming geometry) and 190 x 190 (577 KB: small-vector geometry): the loop enqueues two iterations per sweep like the
    eigen-solver's; exp(-i dt H) v, the iteration count and the norm against the oracle's Exponentiator::run with the same flag."""
    n = side * side
    csr = G.torus_np(side)
    v = G.start_vector(n, 1, np.complex128)
    op = L.CsrOperator(ctx, *csr)
    eng = L.Exponentiator(op, n)
    eng.full_orthogonalize = True
    out, itern = eng.run(-1j * dt, v)
    st = eng.last_stats
    o_out, o_it, _ = oracle.expo(csr, -1j * dt, v, full_orthogonalize=True)
    assert abs(itern - o_it) <= 1, (itern, o_it)
    assert st["pair_iterations"] >= itern - 4, st
    nv = np.linalg.norm(v)
    assert np.max(np.abs(out - o_out)) <= 1e-10 * nv
    assert 1.0 - abs(np.vdot(o_out, out)) / (np.linalg.norm(o_out) * np.linalg.norm(out)) <= 10 * np.finfo(float).eps + 1e-15
    assert abs(np.linalg.norm(out) / nv - 1.0) <= 1e-12
    # and the switch: LL_PAIR_GS=0 keeps the one-sweep form, same output to rounding
    ctx.set_tuning("pair_gs", "0")
    try:
        out1, it1 = eng.run(-1j * dt, v)
    finally:
        ctx.set_tuning("pair_gs", None)
    assert eng.last_stats["pair_iterations"] == 0 and it1 == itern
    assert np.max(np.abs(out - out1)) <= 1e-12 * nv
    op.close()


def test_run_iteration_with_eigenvectors_in_orthogonalize_to_takes_the_one_sweep_forms(ctx, oracle):
    """LambdaLanczos::run_iteration with a caller's orthogonalizeTo list (LL:216-220,259).  The list carries no eigenvalues, so the
    one-sweep / pair forms — whose compensation needs the image of every column under the operator — used to be off for such passes.
    Now the Rayleigh quotients theta_i = <z_i, A z_i> and the residuals ||A z_i - theta_i z_i|| are measured at pass start
    (LoopState::begin_pass) and a list of EIGENvectors takes the locked-column path: the two largest pairs of randsym n = 1e5 as the
    list, three further pairs against the oracle's run_iteration.  A list that is orthonormal but NOT made of eigenvectors fails the
    residual gate and keeps the two-sweep form — same answers."""
    n = 100_003
    csr = G.randsym_np(n)
    init = G.start_vector(n, 1)
    op = L.CsrOperator(ctx, *csr)
    top = L.LambdaLanczos(op, n, True, 2)
    top.init_vector = fixed_init(init)
    lam, lock = top.run()
    lock = np.ascontiguousarray(lock)
    eng = L.LambdaLanczos(op, n, True, 1)
    eng.init_vector = fixed_init(G.start_vector(n, 7))
    vals, vecs, itern = eng.run_iteration(3, lock)
    st = eng.last_stats
    ora = oracle.run_iteration(csr, G.start_vector(n, 7), True, 3, orth=lock)
    assert abs(itern - ora["itern"]) <= 2, (itern, ora["itern"])
    assert st["lagged_iterations"] >= itern - 3 and st["pair_iterations"] >= itern - 6 - 4 * st["second_passes"], st
    for got, ref in zip(vals, ora["eigenvalues"]):
        assert abs(got - ref) <= 1e-10 * max(1.0, abs(ref))
    for i in range(3):
        assert 1 - abs(np.vdot(vecs[i], ora["eigenvectors"][i])) <= 1e-8
        assert np.max(np.abs(lock.conj() @ vecs[i])) <= 1e-9          # orthogonal to the list
    assert vals[0] < lam[1] - 1e-6                                     # the list's pairs are not found again
    # ---- an orthonormal list that is not made of eigenvectors: gate -> two-sweep form, still the oracle's answers
    rng = np.random.default_rng(5)
    q, _ = np.linalg.qr(rng.standard_normal((n, 2)))
    junk = np.ascontiguousarray(q.T)
    eng2 = L.LambdaLanczos(op, n, True, 1)
    eng2.init_vector = fixed_init(G.start_vector(n, 7))
    eng2.max_iteration = 60
    v2, w2, it2 = eng2.run_iteration(2, junk)
    assert eng2.last_stats["lagged_iterations"] == 0 and eng2.last_stats["pair_iterations"] == 0, eng2.last_stats
    ora2 = oracle.run_iteration(csr, G.start_vector(n, 7), True, 2, orth=junk, max_iteration=60)
    assert it2 == ora2["itern"] == 60
    for got, ref in zip(v2, ora2["eigenvalues"]):
        assert abs(got - ref) <= 1e-10 * max(1.0, abs(ref))
    op.close()


@pytest.mark.parametrize("name,window,num_eigs", [("randsym", 40, 1), ("randsym", 41, 1), ("randsym", None, 3), ("torus", 37, 1),
                                                   ("torus", 38, 1), ("laplace", 301, 1)])
def test_pending_pair_enters_the_ritz_gemv_through_its_raw_vectors(ctx, oracle, name, window, num_eigs):
    """At the end of a pass the last pair's vectors u_P (and u_{P+1}) exist only as raw vectors with their measured coefficients.
    They are not completed by sweeps of their own any more: the late update is folded into the coefficients of the Ritz GEMV
    (LoopState::PairTail; compute_eigenvectors, LL:33-62).  Windows that end on the first and on the second vector of a pair, runs
    to convergence, restart passes behind locked eigenvectors, real and complex: the eigenvectors equal those of the flush path
    (setting ritz_tail = 0) to rounding and the oracle's to the usual tolerance; eigenvalues and traces are the same bits."""
    if name == "randsym":
        n = 200_003
        csr, init, find_max, offset = G.randsym_np(n), G.start_vector(n, 1), True, 0.0
    elif name == "laplace":
        m = 400
        n = m * m
        csr, init, find_max, offset = G.laplace2d_np(m), G.start_vector(n, 2), True, 0.0
    else:
        m = 300
        n = m * m
        csr, init, find_max, offset = G.torus_np(m), G.start_vector(n, 3, np.complex128), False, -10.0
    op = L.CsrOperator(ctx, *csr)
    got = {}
    for tail in ("1", "0"):
        ctx.set_tuning("ritz_tail", tail)
        try:
            eng = L.LambdaLanczos(op, n, find_max, num_eigs)
            eng.eigenvalue_offset = offset
            eng.init_vector = fixed_init(init)
            if window:
                eng.max_iteration = window
            vals, vecs = eng.run()
            got[tail] = (vals, vecs, eng.getIterationCounts(), eng.last_alpha.copy(), dict(eng.last_stats))
        finally:
            ctx.set_tuning("ritz_tail", None)
    a, b = got["1"], got["0"]
    assert a[4]["pair_iterations"] > 0
    assert a[2] == b[2] and np.array_equal(a[0], b[0]) and np.array_equal(a[3], b[3])
    for va, vb in zip(a[1], b[1]):
        assert abs(abs(np.vdot(va, vb)) - 1.0) <= 1e-13 and np.max(np.abs(va - vb * np.vdot(vb, va))) <= 1e-12
    ora = oracle.lanczos(csr, init, find_max, num_eigs=num_eigs, offset=offset, max_iteration=window)
    assert a[2] == ora["iter_counts"] or all(abs(x - y) <= 2 for x, y in zip(a[2], ora["iter_counts"]))
    for i in range(num_eigs):
        assert abs(a[0][i] - ora["eigenvalues"][i]) <= 1e-10 * max(1.0, abs(ora["eigenvalues"][i] + offset))
        assert 1 - abs(np.vdot(a[1][i], ora["eigenvectors"][i])) <= 1e-8
    op.close()
