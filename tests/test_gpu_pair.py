"""The pair form of the Gram-Schmidt step — TWO Lanczos iterations per sweep over the basis (kernels.hip "pair" section,
LoopState::enqueue_pair; tools/pair_gs_model.py is the executable specification) — against the reference's sequential
modified Gram-Schmidt (LL:260 -> LA:132-144) through the oracle, and against the two forms it replaces on the same operator:
    LL_FUSE_LAUNCHES=1   two sweeps per iteration (multi-dot, multi-axpy)
    LL_PAIR_GS=0         one sweep per iteration (lagged, compensated)
    default              one sweep per TWO iterations wherever the vectors take the streaming geometry
Tolerances as everywhere: alpha / beta 1e-10 ||A||_inf against the oracle (1e-11 between the device forms),
iteration counts equal, eigenvalue 1e-10, eigenvector 1 - overlap <= 1e-8.  The long fixtures from the REAL reference that
run the pair form in the default geometry are in tests/test_gpu_long_runs.py (laplace400: 1448 iterations, torus300: 524)."""
import numpy as np
import pytest

import lambda_lanczos_amd as L
from lambda_lanczos_amd import generators as G
from util import inf_norm, overlap

pytestmark = pytest.mark.gpu


def fixed_init(v):
    return lambda out, *_: np.copyto(out, v)


def _case(name):
    if name == "randsym":
        n = 30011
        return n, G.randsym_np(n), G.start_vector(n, 1), True, 0.0
    if name == "laplace":
        m = 173
        return m * m, G.laplace2d_np(m), G.start_vector(m * m, 2), True, 0.0
    N = 160
    return N * N, G.torus_np(N), G.start_vector(N * N, 3, np.complex128), False, -10.0


def _run(ctx, op, n, find_max, offset, init, num_eigs=1, max_iteration=None):
    eng = L.LambdaLanczos(op, n, find_max, num_eigs)
    eng.eigenvalue_offset = offset
    eng.init_vector = fixed_init(init)
    if max_iteration:
        eng.max_iteration = max_iteration
    vals, vecs = eng.run()
    return dict(vals=vals, vecs=vecs, iters=eng.getIterationCounts(), alpha=eng.last_alpha.copy(), beta=eng.last_beta.copy(),
                stats=dict(eng.last_stats))


@pytest.mark.parametrize("name", ["randsym", "laplace", "torus"])
def test_pair_form_against_the_oracle_and_the_forms_it_replaces(ctx, oracle, llenv, name):
    """Whole runs to convergence (245 to several hundred iterations; real and complex), streaming geometry forced on all three
    forms.  The pair form must take (almost) every iteration, reproduce the oracle's alpha / beta / count / eigenpair, and agree
    with the one-sweep and the two-sweep form to 1e-11 ||A||."""
    n, csr, init, find_max, offset = _case(name)
    llenv.setenv("LL_BLAS_SMALL_BYTES", "0")
    op = L.CsrOperator(ctx, *csr)
    got = {}
    for form, env in (("two_sweep", {"LL_FUSE_LAUNCHES": "1"}), ("one_sweep", {"LL_PAIR_GS": "0"}), ("pair", {})):
        for k, v in env.items():
            llenv.setenv(k, v)
        got[form] = _run(ctx, op, n, find_max, offset, init)
        for k in env:
            llenv.delenv(k)
    pair, one, two = got["pair"], got["one_sweep"], got["two_sweep"]
    itern = pair["iters"][0]
    assert two["stats"]["lagged_iterations"] == 0 and two["stats"]["pair_iterations"] == 0
    assert one["stats"]["pair_iterations"] == 0 and one["stats"]["lagged_iterations"] >= one["iters"][0] - 3
    # iterations 1 and 2 set the pipeline up; a DGKS repair costs the pair it hits and the two iterations that re-enter
    assert pair["stats"]["pair_iterations"] >= itern - 3 - 4 * pair["stats"]["second_passes"], pair["stats"]
    assert pair["iters"] == one["iters"] == two["iters"]
    scale = inf_norm(csr) + abs(offset)
    for other in (one, two):
        assert np.max(np.abs(pair["alpha"] - other["alpha"])) <= 1e-11 * scale
        assert np.max(np.abs(pair["beta"] - other["beta"])) <= 1e-11 * scale
        assert abs(pair["vals"][0] - other["vals"][0]) <= 1e-12 * scale
        assert 1 - overlap(pair["vecs"][0], other["vecs"][0]) <= 1e-10
    ora = oracle.lanczos(csr, init, find_max, offset=offset)
    assert pair["iters"] == ora["iter_counts"]
    m = len(ora["alpha"])
    assert np.max(np.abs(pair["alpha"][:m] - ora["alpha"])) <= 1e-10 * scale
    assert np.max(np.abs(pair["beta"][:m - 1] - ora["beta"][:m - 1])) <= 1e-10 * scale
    assert abs(pair["vals"][0] - ora["eigenvalues"][0]) <= 1e-10 * max(1.0, abs(pair["vals"][0] + offset))
    assert 1 - overlap(pair["vecs"][0], ora["eigenvectors"][0]) <= 1e-8
    op.close()


@pytest.mark.parametrize("window", [1, 2, 3, 4, 5, 6, 7, 40, 41])
def test_pair_form_with_every_parity_of_the_iteration_count(ctx, oracle, llenv, window):
    """max_iteration = 1 .. 7, 40, 41: windows that end on the first or on the second iteration of a pair, windows too short to
    enter the form at all — the returned pair, the traces and the count against the oracle.  A pair never runs past
    max_iteration (round 6: an odd last iteration takes the one-sweep form instead of a pair whose second half nobody asked for)."""
    n = 30011
    csr, init = G.randsym_np(n), G.start_vector(n, 1)
    llenv.setenv("LL_BLAS_SMALL_BYTES", "0")
    op = L.CsrOperator(ctx, *csr)
    r = _run(ctx, op, n, True, 0.0, init, max_iteration=window)
    ora = oracle.lanczos(csr, init, True, max_iteration=window)
    assert r["iters"] == ora["iter_counts"] == [window]
    assert len(r["alpha"]) == window
    assert np.max(np.abs(r["alpha"] - ora["alpha"][:window])) <= 1e-10 * 30
    if window > 1:
        assert np.max(np.abs(r["beta"][:window - 1] - ora["beta"][:window - 1])) <= 1e-10 * 30
    assert abs(r["vals"][0] - ora["eigenvalues"][0]) <= 1e-10 * abs(r["vals"][0])
    assert 1 - overlap(r["vecs"][0], ora["eigenvectors"][0]) <= 1e-8
    assert r["stats"]["pair_iterations"] == (0 if window < 3 else 2 * ((window - 2) // 2))
    op.close()


def test_pair_form_with_a_second_gram_schmidt_pass_in_every_iteration(ctx, oracle, llenv):
    """LL_DGKS_THRESHOLD=2 makes the host ask for the second pass after EVERY iteration: each pair is cut short by the repair
    of its first vector (flush of the pending late updates, second pass on the completed vector, re-enqueue from a clean state)
    — the results must not change."""
    n = 30011
    csr, init = G.randsym_np(n), G.start_vector(n, 1)
    llenv.setenv("LL_BLAS_SMALL_BYTES", "0")
    llenv.setenv("LL_DGKS_THRESHOLD", "2.0")
    op = L.CsrOperator(ctx, *csr)
    r = _run(ctx, op, n, True, 0.0, init, max_iteration=40)
    ora = oracle.lanczos(csr, init, True, max_iteration=40)
    assert r["iters"] == ora["iter_counts"] == [40] and r["stats"]["second_passes"] >= 38
    assert np.max(np.abs(r["alpha"] - ora["alpha"][:40])) <= 1e-10 * 30
    assert np.max(np.abs(r["beta"][:39] - ora["beta"][:39])) <= 1e-10 * 30
    assert abs(r["vals"][0] - ora["eigenvalues"][0]) <= 1e-10 * abs(r["vals"][0])
    assert 1 - overlap(r["vecs"][0], ora["eigenvectors"][0]) <= 1e-8
    op.close()


def test_pair_form_leaves_through_its_gate_when_the_krylov_space_is_exhausted(ctx, oracle, llenv):
    """An operator with 5 distinct eigenvalues exhausts its Krylov space after 5 iterations: the first vector of the pair (5, 6)
    is rounding noise (beta ~ 1e-15) whose components along the stored vectors are NOT small against its norm.  The fold's gate
    (kPairGate) must catch it: iteration 5 stands (its coefficients were measured), iteration 6 — which took the noise vector as
    its operator input — is done again in the one-sweep form (exact for coefficients of any size), which the rest of the pass
    keeps.  Same iteration count and eigenpair as the oracle's sequential MGS."""
    rng = np.random.default_rng(4)
    n = 300
    q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    lam = np.repeat([1.0, 2.0, 3.5, 5.0, 9.0], n // 5)
    a = (q * lam) @ q.T
    a = (a + a.T) / 2
    init = G.start_vector(n, 1)
    llenv.setenv("LL_BLAS_SMALL_BYTES", "0")
    op = L.DenseOperator(ctx, a)
    eng = L.LambdaLanczos(op, n, True, 1)
    eng.init_vector = fixed_init(init)
    vals, vecs = eng.run()
    ora = oracle.lanczos(G.dense_to_csr(a), init, True)
    assert abs(vals[0] - 9.0) <= 1e-10 and abs(ora["eigenvalues"][0] - 9.0) <= 1e-10
    assert abs(eng.getIterationCounts()[0] - ora["iter_counts"][0]) <= 1
    assert np.linalg.norm(a @ vecs[0] - vals[0] * vecs[0]) <= 1e-9
    st = eng.last_stats
    assert st["pair_iterations"] >= 2 and st["pair_gate_trips"] == 1, st
    m = 4   # the recurrence up to the exhaustion
    assert np.max(np.abs(eng.last_alpha[:m] - ora["alpha"][:m])) <= 1e-10 * 9 and np.max(np.abs(eng.last_beta[:m] - ora["beta"][:m])) <= 1e-10 * 9
    op.close()


def test_pair_form_on_a_ring_whose_krylov_space_is_exhausted_at_the_end(ctx, oracle, llenv):
    """A ring of 400 sites has 201 distinct eigenvalues: the run ends where the Krylov space is exhausted (beta collapses to
    ~1e-13 at m = 201): eigenvalue, iteration count and the traces up to the exhaustion against the oracle, in the pair form."""
    n = 400
    csr, init = G.ring_csr(n), G.start_vector(n, 1)
    llenv.setenv("LL_BLAS_SMALL_BYTES", "0")
    op = L.CsrOperator(ctx, *csr)
    eng = L.LambdaLanczos(op, n, False, 1)
    eng.eigenvalue_offset = -3.0
    eng.init_vector = fixed_init(init)
    vals, vecs = eng.run()
    ora = oracle.lanczos(csr, init, False, offset=-3.0)
    assert abs(eng.getIterationCounts()[0] - ora["iter_counts"][0]) <= 2
    assert abs(vals[0] - ora["eigenvalues"][0]) <= 1e-10 * 5
    m = min(195, len(ora["alpha"]), len(eng.last_alpha))
    assert np.max(np.abs(eng.last_alpha[:m] - ora["alpha"][:m])) <= 1e-10 * 5
    assert np.max(np.abs(eng.last_beta[:m] - ora["beta"][:m])) <= 1e-10 * 5
    assert eng.last_stats["pair_iterations"] >= 190
    op.close()


@pytest.mark.parametrize("dtype", [np.float32, np.complex64], ids=["s", "c"])
def test_pair_form_in_single_precision(ctx, llenv, dtype):
    """float / complex<float> storage (reductions in double): pair form against the two-sweep form on the same operator."""
    wide = np.complex128 if dtype == np.complex64 else np.float64
    n = 30011
    base = G.randsym_np(n)
    csr = (base[0], base[1], base[2].astype(dtype))
    init = G.start_vector(n, 1, wide).astype(dtype)
    llenv.setenv("LL_BLAS_SMALL_BYTES", "0")
    op = L.CsrOperator(ctx, *csr)
    got = {}
    for fuse in ("1", "2"):
        llenv.setenv("LL_FUSE_LAUNCHES", fuse)
        got[fuse] = _run(ctx, op, n, True, 0.0, init, num_eigs=1)
    two, pair = got["1"], got["2"]
    assert pair["stats"]["pair_iterations"] >= pair["iters"][0] - 3 - 4 * pair["stats"]["second_passes"]
    assert abs(pair["iters"][0] - two["iters"][0]) <= 2
    assert np.max(np.abs(pair["alpha"][:12] - two["alpha"][:12])) <= 2e-4 * 30
    assert abs(pair["vals"][0] - two["vals"][0]) <= 2e-3 * 30
    assert 1 - overlap(pair["vecs"][0].astype(wide), two["vecs"][0].astype(wide)) <= 1e-4
    op.close()


@pytest.mark.parametrize("name", ["randsym", "laplace"])
def test_pair_form_in_restart_passes_with_locked_eigenvectors(ctx, oracle, llenv, name):
    """Three eigenpairs (LL:334-354): every pass after the first orthogonalises against the locked eigenvectors (LL:233,259).
    Their columns take lambda_i c_i in the pair form's prediction and quadratic forms (tools/pair_gs_model.py, locked > 0), so
    the restart passes run two iterations per sweep too wherever the locked residuals pass the one-sweep form's gate (the well
    separated top of the random matrix's spectrum).  Same passes, counts and eigenpairs as the one-sweep form and the oracle."""
    n, csr, init, find_max, offset = _case(name)
    llenv.setenv("LL_BLAS_SMALL_BYTES", "0")
    op = L.CsrOperator(ctx, *csr)
    llenv.setenv("LL_PAIR_GS", "0")
    one = _run(ctx, op, n, find_max, offset, init, num_eigs=3)
    llenv.delenv("LL_PAIR_GS")
    pair = _run(ctx, op, n, find_max, offset, init, num_eigs=3)
    assert len(pair["iters"]) >= 2 and pair["iters"] == one["iters"]      # at least one pass behind locked vectors
    assert one["stats"]["pair_iterations"] == 0
    if name == "randsym":   # every pass in the pair form (set-up iterations and DGKS repairs aside)
        assert pair["stats"]["pair_iterations"] >= sum(pair["iters"]) - 3 * len(pair["iters"]) - 4 * pair["stats"]["second_passes"], \
            pair["stats"]
    else:                   # at least the first pass
        assert pair["stats"]["pair_iterations"] >= pair["iters"][0] - 3 - 4 * pair["stats"]["second_passes"], pair["stats"]
    scale = inf_norm(csr) + abs(offset)
    ora = oracle.lanczos(csr, init, find_max, num_eigs=3, offset=offset)
    assert pair["iters"] == ora["iter_counts"]
    assert np.max(np.abs(pair["vals"] - one["vals"])) <= 1e-11 * scale
    assert np.max(np.abs(pair["vals"] - ora["eigenvalues"])) <= 1e-10 * scale
    for i in range(3):
        assert 1 - overlap(pair["vecs"][i], ora["eigenvectors"][i]) <= 1e-8
        for j in range(i):
            assert abs(np.vdot(pair["vecs"][i], pair["vecs"][j])) <= 1e-9
    op.close()


@pytest.mark.parametrize("name,split", [("randsym", 37), ("laplace", 64), ("torus", 50), ("laplace", 1)])
def test_split_sweeps_change_no_bit(ctx, llenv, name, split):
    """One workgroup of the pair sweep keeps 4 x (2 R K + 5 R + 1) partial columns in LDS: beyond K = 2 497 stored vectors (1 247
    complex) the sweep is split into launches over consecutive groups of the stored vectors, the three running strips handed over
    through memory (kernels.hip pair_sweep_kernel).  Every coefficient column is summed in exactly one launch over the same strips
    by the same waves, a strip written and read back is the same bits: with the group size forced down (LL_TEST_PAIR_SPLIT: ragged
    groups, one vector per launch) whole runs — restart passes behind locked vectors included — reproduce the unsplit run bit for
    bit: alpha, beta, counts, eigenvalues, eigenvectors."""
    n, csr, init, find_max, offset = _case(name)
    llenv.setenv("LL_BLAS_SMALL_BYTES", "0")
    op = L.CsrOperator(ctx, *csr)
    num_eigs = 2 if name == "randsym" else 1
    cap = 60 if split == 1 else None          # one launch per stored vector: a bounded window
    whole = _run(ctx, op, n, find_max, offset, init, num_eigs=num_eigs, max_iteration=cap)
    llenv.setenv("LL_TEST_PAIR_SPLIT", str(split))
    parts = _run(ctx, op, n, find_max, offset, init, num_eigs=num_eigs, max_iteration=cap)
    llenv.delenv("LL_TEST_PAIR_SPLIT")
    assert whole["iters"] == parts["iters"]
    # (the counts themselves include speculative pairs that were dropped at the end of a pass: they depend on how far the host ran
    # ahead and differ from run to run)
    assert whole["stats"]["pair_iterations"] > 0 and parts["stats"]["pair_iterations"] > 0
    assert np.array_equal(whole["alpha"], parts["alpha"]) and np.array_equal(whole["beta"], parts["beta"])
    assert np.array_equal(whole["vals"], parts["vals"])
    for a, b in zip(whole["vecs"], parts["vecs"]):
        assert np.array_equal(a, b)
    op.close()


@pytest.mark.parametrize("name,split", [("randsym", None), ("laplace", 37), ("torus", None), ("torus", 51)])
def test_software_pipelined_sweep_changes_no_bit(ctx, name, split):
    """The production sweep (pair_sweep_pipe_kernel: the next trip's strips requested before the current trip is consumed, stored
    vectors addressed through a device pointer table, two vectors per trip, trips across slab boundaries, dummy columns with zero
    coefficients behind the last stored vector; the default on vectors of more than ~9 MiB) performs the additions of the reference
    kernel (pair_sweep_kernel, setting sweep_pipeline = 0: four vectors per trip over the segment lists) in the same order: whole runs — restart passes behind locked
    vectors, ragged last strips, split sweeps included — agree bit for bit: alpha, beta, counts, eigenvalues, eigenvectors."""
    n, csr, init, find_max, offset = _case(name)
    ctx.set_tuning("blas_small_bytes", "0")
    if split:
        ctx.set_tuning("pair_split", str(split))
    try:
        op = L.CsrOperator(ctx, *csr)
        num_eigs = 2 if name == "randsym" else 1
        ctx.set_tuning("sweep_pipeline", "2")      # (by itself only on vectors of more than ~9 MiB: forced onto these small cases)
        piped = _run(ctx, op, n, find_max, offset, init, num_eigs=num_eigs)
        ctx.set_tuning("sweep_pipeline", "0")
        plain = _run(ctx, op, n, find_max, offset, init, num_eigs=num_eigs)
    finally:
        for key in ("sweep_pipeline", "pair_split", "blas_small_bytes"):
            ctx.set_tuning(key, None)
    assert piped["iters"] == plain["iters"]
    assert piped["stats"]["pair_iterations"] > 0 and plain["stats"]["pair_iterations"] > 0
    assert np.array_equal(piped["alpha"], plain["alpha"]) and np.array_equal(piped["beta"], plain["beta"])
    assert np.array_equal(piped["vals"], plain["vals"])
    for a, b in zip(piped["vecs"], plain["vecs"]):
        assert np.array_equal(a, b)
    op.close()


@pytest.mark.parametrize("name,n", [("randsym", 70001), ("randsym", 100003), ("laplace", 260), ("torus", 190)])
def test_pair_form_in_the_small_vector_geometry(ctx, oracle, llenv, name, n):
    """Vectors of 512 KiB .. 1 MiB (n = 6.6e4 .. 1.3e5 doubles — the reference's everyday sizes) take the pair form in the small-vector
    geometry (pair_small_kernel: four waves per 1 KiB strip split the stored vectors; round 6).  Default switches, whole runs to
    convergence with two or three roots (restart passes behind locked eigenvectors included): alpha / beta / counts / eigenpairs
    against the oracle's sequential MGS, and against the one-sweep form of the same geometry (LL_PAIR_GS=0) to 1e-11 ||A||."""
    if name == "randsym":
        csr, init, find_max, offset, dim = G.randsym_np(n), G.start_vector(n, 1), True, 0.0, n
    elif name == "laplace":
        csr, init, find_max, offset, dim = G.laplace2d_np(n), G.start_vector(n * n, 2), True, 0.0, n * n
    else:
        csr, init, find_max, offset, dim = G.torus_np(n), G.start_vector(n * n, 3, np.complex128), False, -10.0, n * n
    assert (512 << 10) <= dim * csr[2].dtype.itemsize < (1 << 20)
    op = L.CsrOperator(ctx, *csr)
    num_eigs = 2 if n == 70001 else 1
    pair = _run(ctx, op, dim, find_max, offset, init, num_eigs=num_eigs)
    llenv.setenv("LL_PAIR_GS", "0")
    one = _run(ctx, op, dim, find_max, offset, init, num_eigs=num_eigs)
    llenv.delenv("LL_PAIR_GS")
    ora = oracle.lanczos(csr, init, find_max, num_eigs=num_eigs, offset=offset)
    norm = inf_norm(csr) + abs(offset)
    assert one["stats"]["pair_iterations"] == 0
    total = sum(pair["iters"])
    assert pair["stats"]["pair_iterations"] >= total - 3 * len(pair["iters"]) - 4 * pair["stats"]["second_passes"] - 2 * pair["stats"]["pair_gate_trips"], pair["stats"]
    assert len(pair["iters"]) == len(ora["iter_counts"]) == len(one["iters"])
    for a, b, c in zip(pair["iters"], ora["iter_counts"], one["iters"]):
        assert abs(a - b) <= 2 and abs(c - b) <= 2, (pair["iters"], ora["iter_counts"], one["iters"])
    m = min(len(pair["alpha"]), len(one["alpha"]), 200)
    if len(pair["iters"]) == 1:   # (the trace of the LAST pass: comparable with the oracle's when there is one pass)
        mo = min(m, len(ora["alpha"]))
        assert np.max(np.abs(pair["alpha"][:mo] - ora["alpha"][:mo])) <= 1e-10 * norm
        assert np.max(np.abs(pair["beta"][:mo - 1] - ora["beta"][:mo - 1])) <= 1e-10 * norm
        assert np.max(np.abs(pair["alpha"][:m] - one["alpha"][:m])) <= 1e-11 * norm
    for i in range(num_eigs):
        assert abs(pair["vals"][i] - ora["eigenvalues"][i]) <= 1e-10 * max(1.0, abs(ora["eigenvalues"][i] + offset))
        assert 1 - overlap(pair["vecs"][i], ora["eigenvectors"][i]) <= 1e-8
    op.close()


@pytest.mark.parametrize("name,limit", [("laplace", 100), ("laplace", 101), ("torus", 77)])
def test_pair_form_hands_over_to_the_one_sweep_form_at_its_column_limit(ctx, oracle, llenv, name, limit):
    """The coefficient records of the pair form end at 4 992 real (2 492 complex) stored vectors; beyond that the loop completes the
    pending pair and continues with one sweep per iteration.  No test problem runs that long, so LL_TEST_PAIR_MAX_STORED moves the
    hand-over into reach (even and odd limits: either vector of a pair can be the last one): same run as the pair form alone to
    1e-11 ||A||, the oracle's counts and eigenpair, and both forms really ran."""
    n, csr, init, find_max, offset = _case(name)
    llenv.setenv("LL_BLAS_SMALL_BYTES", "0")
    op = L.CsrOperator(ctx, *csr)
    pair = _run(ctx, op, n, find_max, offset, init)
    llenv.setenv("LL_TEST_PAIR_MAX_STORED", str(limit))
    both = _run(ctx, op, n, find_max, offset, init)
    llenv.delenv("LL_TEST_PAIR_MAX_STORED")
    itern = both["iters"][0]
    assert both["iters"] == pair["iters"] and itern > limit + 50
    assert limit - 4 <= both["stats"]["pair_iterations"] <= limit + 6, both["stats"]
    assert both["stats"]["lagged_iterations"] >= itern - 3 - 4 * both["stats"]["second_passes"], both["stats"]
    scale = inf_norm(csr) + abs(offset)
    assert np.max(np.abs(both["alpha"] - pair["alpha"])) <= 1e-11 * scale
    assert np.max(np.abs(both["beta"] - pair["beta"])) <= 1e-11 * scale
    assert abs(both["vals"][0] - pair["vals"][0]) <= 1e-12 * scale
    assert 1 - overlap(both["vecs"][0], pair["vecs"][0]) <= 1e-10
    ora = oracle.lanczos(csr, init, find_max, offset=offset)
    assert both["iters"] == ora["iter_counts"]
    assert abs(both["vals"][0] - ora["eigenvalues"][0]) <= 1e-10 * max(1.0, abs(both["vals"][0] + offset))
    assert 1 - overlap(both["vecs"][0], ora["eigenvectors"][0]) <= 1e-8
    op.close()
