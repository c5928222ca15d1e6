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
