"""The reference's own test problems (test/lambda_lanczos_test.cpp = T1, test/exponentiator_test.cpp = T2) and the
SURVEY 8(d) generators, as data: CSR operator + engine settings + known answer.  Shared by the oracle tests (CPU),
the HIP parity tests (GPU) and tests/golden/make_golden.py."""
import math

import numpy as np

from lambda_lanczos_amd import generators as G

EPS = float(np.finfo(np.float64).eps)

M3 = np.array([[2.0, 1.0, 1.0], [1.0, 2.0, 1.0], [1.0, 1.0, 2.0]])                     # T1:130
H3 = np.array([[0, 1j, 1], [-1j, 0, 1j], [1, -1j, 0]], dtype=np.complex128)            # T1:378
M8 = np.array([[6, -3, -3, 0, -1, 1, -1, 1], [-3, -4, 2, 2, -1, -5, 0, -4], [-3, 2, 2, -3, 0, 0, -1, -1],
               [0, 2, -3, 0, -3, 3, 2, 2], [-1, -1, 0, -3, -2, 0, -5, -4], [1, -5, 0, 3, 0, -4, 5, 0],
               [-1, 0, -1, 2, -5, 5, -4, 4], [1, -4, -1, 2, -4, 0, 4, 2]], dtype=np.float64)  # T1:446-453
M8_VALS = [-13.21508597, -8.50033154, -4.26674892]                                      # T1:472
M8_VECS = np.array([                                                                    # T1:473-476
    [0.02081752, -0.49222707, 0.13202088, 0.24048092, 0.15089223, -0.60850056, 0.48079787, -0.24043829],
    [0.16645991, 0.51818471, -0.00646562, -0.09493495, 0.60595718, 0.02042567, 0.52346924, 0.23043415],
    [0.03381669, -0.07999997, 0.32090331, 0.61650970, 0.41812886, -0.01782613, -0.45571810, 0.35575946]])


def eigen_cases():
    """name -> dict(csr, dtype, find_maximum, settings..., expect...) for the known-answer eigen tests."""
    c = {}
    c["simple_matrix"] = dict(csr=G.dense_to_csr(M3), find_maximum=True, num_eigs=1, offset=6.0, eps=None,
                              values=[4.0], vectors=[np.ones(3) / math.sqrt(3)], ref="T1:128-161")
    c["dynamic_matrix"] = dict(csr=G.chain_csr(10), find_maximum=False, num_eigs=1, offset=-10.0, eps=1e-14,
                               values=[-2.0 * math.cos(math.pi / 11)],
                               vectors=[np.sin((np.arange(10) + 1) * math.pi / 11) / np.linalg.norm(
                                   np.sin((np.arange(10) + 1) * math.pi / 11))], ref="T1:262-308")
    c["simple_matrix_complex"] = dict(csr=G.dense_to_csr(M3.astype(np.complex128)), find_maximum=True, num_eigs=1,
                                      offset=0.0, eps=None, values=[4.0],
                                      vectors=[np.ones(3, dtype=np.complex128) / math.sqrt(3)], ref="T1:310-343")
    c["hermitian_matrix"] = dict(csr=G.dense_to_csr(H3), find_maximum=False, num_eigs=1, offset=0.0, eps=None,
                                 values=[-2.0], vectors=[np.array([1, 1j, -1]) / math.sqrt(3)], ref="T1:375-409")
    c["single_element"] = dict(csr=G.dense_to_csr(np.array([[2.0]])), find_maximum=True, num_eigs=1, offset=0.0,
                               eps=None, values=[2.0], vectors=[np.array([1.0])], ref="T1:411-440")
    c["multiple_eigenpairs"] = dict(csr=G.dense_to_csr(M8), find_maximum=False, num_eigs=3, offset=0.0, eps=1e-7,
                                    values=M8_VALS, vectors=list(M8_VECS), ref="T1:442-488")
    ring = sorted(-2.0 * math.cos(2.0 * math.pi * j / 50) for j in range(-13, 13))
    c["multiple_degenerate"] = dict(csr=G.ring_csr(50), find_maximum=False, num_eigs=26, offset=0.0, eps=1e-14,
                                    values=ring, vectors=None, ref="T1:490-536", abs_tol=1e-13)
    return c


def plane_wave_exact(n, t, a, inp):
    """exp(a*H) inp for the periodic chain with hopping t, through its analytic plane waves (T2:83-104)."""
    k = 2 * math.pi / n * np.arange(n)
    ev = 2 * t * np.cos(k)
    u = np.exp(1j * np.outer(np.arange(n), k)) / math.sqrt(n)
    return u @ (np.exp(a * ev) * (u.conj().T @ inp))


def expo_cases():
    c = {}
    inp3 = np.array([1.0, 0.0, 0.0])
    w, v = np.linalg.eigh(M3)
    c["exponentiate_real"] = dict(csr=G.dense_to_csr(M3), a=3.0, input=inp3, full=False,
                                  exact=v @ (np.exp(3.0 * w) * (v.T @ inp3)), ref="T2:31-81")
    n = 100
    inp = np.zeros(n, dtype=np.complex128)
    inp[0], inp[n - 1], inp[n // 2] = 1 + 2j, 1 + 2j, 8 + 2j                             # T2:126-130
    inp = inp / np.linalg.norm(inp)
    c["exponentiate_large"] = dict(csr=G.ring_csr(n, -1.0, np.complex128), a=3j, input=inp, full=False,
                                   exact=plane_wave_exact(n, -1.0, 3j, inp), ref="T2:106-162")
    c["exponentiate_zero"] = dict(csr=G.ring_csr(n, -1.0, np.complex128), a=0j, input=inp, full=True,
                                  exact=inp.copy(), ref="T2:164-222")
    return c


def run_eigen_case(engine_cls, make_op, case, init=None):
    """Drive an engine with the reference's public-field idiom (T1:142-147)."""
    csr = case["csr"]
    n = csr[0].shape[0] - 1
    eng = engine_cls(make_op(csr), n, case["find_maximum"], case["num_eigs"])
    if case.get("eps") is not None:
        eng.eps = case["eps"]
    eng.eigenvalue_offset = case["offset"]
    if init is not None:
        eng.init_vector = init
    return eng


def run_iteration_problem(name):
    """(csr, init, dtype) of a tests/golden/run_iteration.json case; orthogonalizeTo comes from the fixture itself."""
    if name.startswith("m8"):
        return G.dense_to_csr(M8), G.start_vector(8, 1)
    if name.startswith("randsym600"):
        return G.randsym_np(600), G.start_vector(600, 1)
    if name.startswith("torus12"):
        return G.torus_np(12), G.start_vector(144, 1, np.complex128)
    raise KeyError(name)
