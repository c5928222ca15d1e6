"""The algebra of the one-sweep (lagged, compensated) Gram-Schmidt form (kernels.hip: lagged_kernel,
lagged_fold_kernel) in numpy, at a size the CPU suite runs in seconds: the statements the design rests on.  The GPU
kernels are checked against the two-sweep form and the oracle in tests/test_gpu_round3.py; this file pins the scheme
itself (tools/lagged_gs_model.py is the same model at the size quoted in DESIGN.md)."""
import importlib.util
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def model():
    spec = importlib.util.spec_from_file_location("lagged_gs_model", os.path.join(ROOT, "tools", "lagged_gs_model.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    # a smaller instance of the same problem
    import scipy.sparse as sp
    m.n, m.K = 1500, 120
    rng = np.random.default_rng(1)
    a = sp.random(m.n, m.n, density=8 / m.n, random_state=3, format="csr")
    m.A = ((a + a.T) * 0.5 + sp.diags(np.linspace(2, 12, m.n))).tocsr()
    m.v0 = rng.uniform(-1, 1, m.n)
    m.v0 /= np.linalg.norm(m.v0)
    m.ref = m.reference()
    return m


def test_without_compensation_the_late_coefficients_grow_geometrically(model):
    out = model.lagged(False, False)
    mc = out[1] if len(out) == 2 else out[3]
    assert mc[9] < 1e-13 and mc[49] > 1e-6 and mc[49] / mc[29] > 1e3   # ~2x per iteration


def test_full_scheme_reproduces_the_recurrence_of_full_reorthogonalisation(model):
    ra, rb, _ = model.ref
    a, b, U, mc = model.lagged(True, True)
    assert np.max(np.abs(a - ra)) <= 1e-12 and np.max(np.abs(b - rb)) <= 1e-12
    assert np.max(mc) <= 1e-13
    K = model.K
    assert np.max(np.abs(U[:K] @ U[:K].T - np.eye(K))) <= 1e-14


@pytest.mark.parametrize("inject", [1e-3, 0.5])
def test_full_scheme_is_exact_for_late_coefficients_of_any_size(model, inject):
    ra, rb, _ = model.ref
    a, b, U, mc = model.lagged(True, True, inject)
    assert abs(np.max(mc) - inject) <= 1e-6 * inject + 1e-12
    assert np.max(np.abs(a - ra)) <= 1e-12 and np.max(np.abs(b - rb)) <= 1e-12


def test_alpha_needs_its_second_order_term(model):
    ra, _, _ = model.ref
    a, _, _, _ = model.lagged(True, False, 1e-3)     # first-order correction of alpha only
    assert 1e-7 <= np.max(np.abs(a - ra)) <= 1e-3   # ~ |c|^2 ||A||
