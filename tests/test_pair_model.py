"""The algebra of the pair form of the Gram-Schmidt step — two Lanczos iterations per sweep over the basis (DESIGN.md 3.2;
kernels.hip "pair" section, LoopState::enqueue_pair) — in numpy, at a size the CPU suite runs in seconds.  tools/pair_gs_model.py
is the executable specification the device kernels were written from (same launches, same formulas, real and complex); the GPU
kernels themselves are checked against the oracle and the real reference in tests/test_gpu_pair.py / test_gpu_long_runs.py."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def model():
    spec = importlib.util.spec_from_file_location("pair_gs_model", os.path.join(ROOT, "tools", "pair_gs_model.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize("complex_", [False, True], ids=["real", "complex"])
def test_pair_form_reproduces_the_recurrence_of_full_reorthogonalisation(model, complex_):
    r = model.measure(complex_, 0.0, n=1500, K=120)
    assert r["iterations"] == 120
    assert r["dalpha"] <= 1e-12 and r["dbeta"] <= 1e-12
    assert r["orth"] <= 1e-14 and r["dvec"] <= 1e-12
    assert r["maxcoef"] <= 1e-12                      # every stored-basis coefficient of an operator input stays eps-sized
    assert r["fold_rho"] <= 1e-13 and r["fold_gam"] <= 1e-13


def test_below_the_gate_planted_components_leave_no_trace(model):
    """Components of relative size 1e-8 (= kPairGate) along stored vectors in one operator input: measured and removed to first
    order, the neglected second-order term is below rounding."""
    r = model.measure(False, 1e-8, n=1500, K=120)
    assert 5e-9 <= r["maxcoef"] <= 5e-8
    assert r["dalpha"] <= 1e-12 and r["dbeta"] <= 1e-12 and r["dvec"] <= 1e-12


def test_beyond_the_gate_one_beta_carries_a_second_order_term_but_the_basis_stays_exact(model):
    """Why the device code has the gate: at 1e-3 one beta is off by ~1e-5 (second order), while the stored vectors and every
    folded quantity are still exact — the fallback to the one-sweep form is about the recurrence coefficients, not the basis."""
    r = model.measure(False, 1e-3, n=1500, K=120)
    assert 1e-8 <= r["dbeta"] <= 1e-3
    assert r["dvec"] <= 1e-12 and r["orth"] <= 1e-14 and r["fold_rho"] <= 1e-13


@pytest.mark.parametrize("complex_,locked", [(False, 3), (True, 2)], ids=["real-3-locked", "complex-2-locked"])
def test_pair_form_behind_locked_eigenvectors(model, complex_, locked):
    """A restart pass (LL:233,259): the locked eigenvectors are the first stored columns of every sweep; their image under the
    operator is lambda_i times the coefficient — in the prediction and in the quadratic forms of the alphas."""
    r = model.measure(complex_, 0.0, n=1200, K=100, locked=locked)
    assert r["dalpha"] <= 1e-12 and r["dbeta"] <= 1e-12 and r["dvec"] <= 1e-12 and r["orth"] <= 1e-13
    assert r["maxcoef"] <= 1e-12
