"""The operator zoo either side of the Krylov loop, through the C ABI on the GPU:
  * the matrix-free lattice operator ll_op_create_stencil_* — the reference's "dynamic matrix" family
    (src/samples/sample3_dynamic.cpp:17-22, T1:262-308 open chain; T2:106-162 periodic ring; BASELINE config 2),
  * the dense row-major operator ll_op_create_dense_* (src/samples/sample1_simple.cpp:22-28; T1:128, T1:442).
Parity: one apply against the oracle's CSR row loop on the equivalent matrix (generators.lattice_csr / dense_to_csr);
whole loops against the oracle, the CSR operator of the same matrix and the reference's known answers."""
import math

import numpy as np
import pytest

import cases
import lambda_lanczos_amd as L
from lambda_lanczos_amd import _capi as capi
from lambda_lanczos_amd import generators as G
from util import overlap

pytestmark = pytest.mark.gpu
EPS = np.finfo(np.float64).eps

LATTICES = {
    # name: (dims, diag, hop, periodic, with_onsite)
    "chain10_open": ([10], 0.0, -1.0, False, False),                      # T1:265-273
    "ring100": ([100], 0.0, -1.0, True, False),                           # T2:113-121
    "single_site_ring": ([1], 0.5, -1.0, True, False),
    "laplace_37x41": ([37, 41], 4.0, -1.0, False, False),                 # BASELINE config 2 at a ragged size
    "torus_300x257": ([300, 257], 0.0, [-1.0, 0.5], True, True),
    "mixed_3d": ([5, 6, 7], 0.25, [0.5, -1.0, 0.75], [True, False, True], True),
    "thin_3d": ([2, 1, 3], 0.0, [1.0, -1.0, 0.5], True, False),           # neighbours reached twice / self neighbours
    "slab_3d": ([40, 33, 29], -1.0, [-1.0, -0.5, -0.25], [False, True, False], False),
    # fastest dimension a multiple of 8: the vectorised kernel (32 bytes of consecutive sites per lane) for every type
    "vec_ring_4096": ([4096], 0.5, -1.0, True, True),
    "vec_chain_64": ([64], 0.0, -1.0, False, False),
    "vec_torus_48x96": ([48, 96], 0.0, [-1.0, 0.5], True, True),
    "vec_open_33x8": ([33, 8], 4.0, -1.0, False, False),                  # one chunk per lattice row (f32)
    "vec_mixed_3d": ([12, 10, 16], 0.25, [0.5, -1.0, 0.75], [True, False, True], True),
    "vec_open_3d": ([9, 7, 24], -1.0, [-1.0, -0.5, -0.25], False, False),
}
COMPLEX_HOPS = {"vec_ring_4096": 0.3 - 1j, "vec_torus_48x96": [-1.0 + 0.2j, 0.5j], "vec_mixed_3d": [0.5 + 1j, -1.0, 0.75j],
                "vec_open_3d": [-1.0, -0.5 + 0.5j, 0.25j], "ring100": 0.3 - 1j, "torus_300x257": [-1.0 + 0.2j, 0.5j], "mixed_3d": [0.5 + 1j, -1.0, 0.75j],
                "thin_3d": [1.0 + 1j, -1j, 0.5], "slab_3d": [-1.0, -0.5 + 0.5j, 0.25j]}


# Peierls phases (complex storage types only): phase_grad[d][e] = d(phase of the hop along d) / d(coordinate e)
PHASES = {"vec_torus_48x96": [[0.0, 0.0], [2 * math.pi * 3 / 48, 0.0]],                 # config 5's Landau gauge
          "torus_300x257": [[0.0, 0.01], [0.02, 0.0]],
          "mixed_3d": [[0.1, 0.2, 0.3], [0.0, 0.4, -0.2], [0.5, 0.0, 0.7]],             # every kind of dependence
          "vec_mixed_3d": [[0.1, 0.2, 0.3], [0.0, 0.4, -0.2], [0.5, 0.0, 0.7]],
          "vec_ring_4096": [[0.003]]}


def lattice(name, dtype):
    dims, diag, hop, periodic, with_onsite = LATTICES[name]
    cplx = np.issubdtype(np.dtype(dtype), np.complexfloating)
    if cplx and name in COMPLEX_HOPS:
        hop = COMPLEX_HOPS[name]
    n = int(np.prod(dims))
    onsite = 0.3 * np.cos(1.7 * np.arange(n)) if with_onsite else None
    kw = dict(diag=diag, hop=hop, periodic=periodic, onsite=onsite, dtype=dtype)
    if cplx and name in PHASES:
        kw["phase_grad"] = PHASES[name]
    return dims, kw


def rnd(n, dtype, seed):
    rng = np.random.default_rng(seed)
    v = rng.uniform(-1, 1, n)
    if np.issubdtype(np.dtype(dtype), np.complexfloating):
        v = v + 1j * rng.uniform(-1, 1, n)
    return v.astype(dtype)


@pytest.mark.parametrize("name", sorted(LATTICES))
@pytest.mark.parametrize("dtype", [np.float64, np.complex128, np.float32, np.complex64])
@pytest.mark.parametrize("offset", [0.0, -2.5])
def test_lattice_apply_matches_oracle(ctx, oracle, name, dtype, offset):
    dims, kw = lattice(name, dtype)
    n = int(np.prod(dims))
    wide = np.complex128 if np.issubdtype(np.dtype(dtype), np.complexfloating) else np.float64
    csr = G.lattice_csr(dims, **dict(kw, dtype=wide))
    single = np.dtype(dtype) in (np.dtype(np.float32), np.dtype(np.complex64))
    if kw["onsite"] is not None and single:
        # the device keeps onsite in the real type of T: compare against the float-rounded values
        csr = G.lattice_csr(dims, **dict(kw, dtype=wide, onsite=kw["onsite"].astype(np.float32).astype(np.float64)))
    op = L.StencilOperator(ctx, dims, **kw)
    assert op.info() == (n, n, 0)
    x = rnd(n, dtype, 11)
    xd, yd = ctx.to_device(x), ctx.empty(n, dtype)
    alpha = L.spmv(op, xd, yd, offset=offset, want_dot=True)
    y = yd.get()
    xw = x.astype(wide)
    y_ref = oracle.spmv(csr, xw) + offset * xw
    tol = 16 * (np.finfo(np.float32).eps if single else EPS)
    scale = op.inf_norm() + abs(offset)
    assert np.max(np.abs(y - y_ref)) <= tol * max(scale, 1.0)
    assert abs(alpha - np.vdot(xw, y_ref).real) <= tol * max(scale, 1.0) * n
    # row-sum bound reported for eigenvalue_offset
    rowsum = np.max(np.add.reduceat(np.abs(csr[2]), csr[0][:-1]))
    assert rowsum <= op.inf_norm() * (1 + (1e-6 if single else 1e-12)) + 1e-12
    op.close()


VEC_CASES = [(nm, dt) for nm in ("vec_ring_4096", "vec_torus_48x96", "vec_mixed_3d", "vec_open_3d")
             for dt in ("float64", "complex128", "float32", "complex64")]


def _apply_all_vec_cases(ctx):
    out = {}
    for nm, dt in VEC_CASES:
        dims, kw = lattice(nm, np.dtype(dt).type)
        n = int(np.prod(dims))
        op = L.StencilOperator(ctx, dims, **kw)
        x = rnd(n, np.dtype(dt).type, 21)
        xd, yd = ctx.to_device(x), ctx.empty(n, x.dtype)
        a = L.spmv(op, xd, yd, offset=0.5, want_dot=True)
        out[nm + "/" + dt] = yd.get()
        out[nm + "/" + dt + "/dot"] = np.array([a])
        op.close()
    return out


def test_vectorised_lattice_kernel_equals_scalar_kernel(ctx, tmp_path):
    """The vectorised kernel adds every site's terms in the order of the one-site-per-lane kernel: identical bits for
    every storage type.  (The scalar run happens in a child process whose context takes the stencil_vec = 0 hook.)"""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = str(tmp_path / "scalar.npz")
    code = ("import sys, numpy as np; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import lambda_lanczos_amd as L; import test_gpu_operators as t; import util; util.install_hook_sync()\n"
            "np.savez(%r, **t._apply_all_vec_cases(L.Context(0)))\n") % (root, os.path.join(root, "tests"), path)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, LL_STENCIL_VEC="0"), capture_output=True, text=True,
                       timeout=200)
    assert r.returncode == 0, r.stdout + r.stderr
    scalar = np.load(path)
    vec = _apply_all_vec_cases(ctx)
    for key, val in vec.items():
        if key.endswith("/dot"):
            assert abs(val[0] - scalar[key][0]) <= 1e-12 * max(1.0, abs(scalar[key][0])), key
        else:
            assert np.array_equal(val, scalar[key]), key


def test_dynamic_matrix_known_answer(ctx):
    """T1:262-308 DYNAMIC_MATRIX through the matrix-free operator: lambda_min = -2 cos(pi/(n+1)), sine eigenvector."""
    n = 10
    op = L.StencilOperator(ctx, [n], diag=0.0, hop=-1.0)
    eng = L.LambdaLanczos(op, n, False, 1)
    eng.eps = 1e-14
    eng.eigenvalue_offset = -10.0
    eng.init_vector = lambda v, *_: v.__setitem__(slice(None), G.start_vector(n, 1))
    vals, vecs = eng.run()
    want = -2.0 * math.cos(math.pi / (n + 1))
    assert abs(vals[0] - want) <= abs(want) * eng.eps * 10
    sine = np.sin((np.arange(n) + 1) * math.pi / (n + 1))
    assert 1 - overlap(vecs[0], sine) <= 1e-13
    op.close()


@pytest.mark.parametrize("name", ["laplace_37x41", "slab_3d", "torus_300x257"])
def test_lattice_lanczos_equals_csr_and_oracle(ctx, oracle, name):
    """Same problem through the lattice operator, the CSR operator and the oracle: same alpha/beta trace, same
    iteration count, same eigenpair."""
    dims, kw = lattice(name, np.float64)
    n = int(np.prod(dims))
    csr = G.lattice_csr(dims, **kw)
    init = G.start_vector(n, 1)
    st = L.StencilOperator(ctx, dims, **kw)
    cs = L.CsrOperator(ctx, *csr)
    runs = []
    for op in (st, cs):
        eng = L.LambdaLanczos(op, n, False, 1)
        eng.eigenvalue_offset = -op.inf_norm()
        eng.max_iteration = 150
        eng.init_vector = lambda v, *_: v.__setitem__(slice(None), init)
        vals, vecs = eng.run()
        runs.append((vals[0], vecs[0], eng.getIterationCounts(), eng.last_alpha, eng.last_beta))
    ora = oracle.lanczos(csr, init, False, offset=-st.inf_norm(), max_iteration=150)
    norm = st.inf_norm()
    for val, vec, iters, alpha, beta in runs:
        assert iters == ora["iter_counts"]
        m = len(ora["alpha"])
        assert np.max(np.abs(alpha[:m] - ora["alpha"])) <= 1e-10 * norm
        assert np.max(np.abs(beta[:m - 1] - ora["beta"][:m - 1])) <= 1e-10 * norm
        assert abs(val - ora["eigenvalues"][0]) <= 1e-10 * max(1.0, norm)
    # the Ritz vector of an unconverged fixed window is still the same vector on all three paths
    assert 1 - overlap(runs[0][1], runs[1][1]) <= 1e-8
    assert 1 - overlap(runs[0][1], ora["eigenvectors"][0]) <= 1e-8
    st.close()
    cs.close()


def test_lattice_exponentiate_large_matrix(ctx):
    """T2:106-162 EXPONENTIATE_LARGE_MATRIX with the ring as a matrix-free operator: analytic plane waves, 19 iterations."""
    case = cases.expo_cases()["exponentiate_large"]
    n = 100
    op = L.StencilOperator(ctx, [n], hop=-1.0, periodic=True, dtype=np.complex128)
    ex = L.Exponentiator(op, n)
    out, itern = ex.run(case["a"], case["input"])
    assert 1 - overlap(out, case["exact"]) <= ex.eps * 10
    assert itern == 19
    tout, terms = ex.taylor_run(case["a"], case["input"])
    assert terms == 37 and 1 - overlap(tout, case["exact"]) <= ex.eps * 10
    op.close()


def test_config5_torus_as_a_lattice_operator(ctx, oracle):
    """BASELINE config 5 (complex Hermitian torus, Landau-gauge Peierls phases on the x hops, random on-site terms)
    expressed matrix-free: the same matrix as generators.torus_np entry by entry, the same exp(-i H dt) v and iteration
    count as the oracle on the CSR form."""
    N = 48
    n = N * N
    csr = G.torus_np(N)
    onsite = G.u01(np.arange(n, dtype=np.uint64)) - 0.5
    kw = dict(diag=0.0, hop=[-1.0, -1.0], periodic=True, onsite=onsite, dtype=np.complex128,
              phase_grad=[[0.0, 0.0], [2 * math.pi * 3 / N, 0.0]])
    as_csr = G.lattice_csr([N, N], **kw)
    import scipy.sparse as sp

    d = sp.csr_matrix((as_csr[2], as_csr[1], as_csr[0]), shape=(n, n)) - sp.csr_matrix((csr[2], csr[1], csr[0]), shape=(n, n))
    assert abs(d).max() <= 1e-15
    op = L.StencilOperator(ctx, [N, N], **kw)
    inp = G.start_vector(n, 1, np.complex128)
    for dt in (0.1, 1.0):
        out, it = L.Exponentiator(op, n).run(-1j * dt, inp)
        o_ref, it_ref, _ = oracle.expo(csr, -1j * dt, inp)
        assert it == it_ref
        assert np.max(np.abs(out - o_ref)) <= 1e-11 * np.linalg.norm(inp)
        assert abs(np.linalg.norm(out) / np.linalg.norm(inp) - 1) <= 1e-12
    op.close()


def test_lattice_rejects_bad_arguments(ctx):
    with pytest.raises(L.LanczosHipError):
        L.StencilOperator(ctx, [4, 4], hop=[1j, 1.0], dtype=np.float64)      # complex hop, real storage
    with pytest.raises(L.LanczosHipError):
        L.StencilOperator(ctx, [4, 4], phase_grad=[[0, 0.1], [0, 0]], dtype=np.float32)   # phases, real storage
    with pytest.raises(L.LanczosHipError):
        L.StencilOperator(ctx, [4, 0])
    with pytest.raises((L.LanczosHipError, ValueError)):
        L.StencilOperator(ctx, [2, 2, 2, 2])
    with pytest.raises(L.LanczosHipError):
        L.StencilOperator(ctx, [8], n_local=4)                               # a single-GPU context needs the whole lattice


# ------------------------------------------------------------------ dense operator
@pytest.mark.parametrize("dtype", [np.float64, np.complex128, np.float32, np.complex64])
@pytest.mark.parametrize("n", [1, 3, 64, 301, 2050])
def test_dense_apply_matches_oracle(ctx, oracle, dtype, n):
    rng = np.random.default_rng(n)
    a = rng.uniform(-1, 1, (n, n))
    if np.issubdtype(np.dtype(dtype), np.complexfloating):
        a = a + 1j * rng.uniform(-1, 1, (n, n))
    a = ((a + a.conj().T) / 2).astype(dtype)
    wide = np.complex128 if np.issubdtype(np.dtype(dtype), np.complexfloating) else np.float64
    op = L.DenseOperator(ctx, a)
    x = rnd(n, dtype, 5)
    xd, yd = ctx.to_device(x), ctx.empty(n, dtype)
    alpha = L.spmv(op, xd, yd, offset=0.75, want_dot=True)
    y_ref = oracle.spmv(G.dense_to_csr(a.astype(wide)), x.astype(wide)) + 0.75 * x.astype(wide)
    tol = 8 * (np.finfo(np.float32).eps if dtype in (np.float32, np.complex64) else EPS)
    assert np.max(np.abs(yd.get() - y_ref)) <= tol * (op.inf_norm() + 1.0)
    assert abs(alpha - np.vdot(x.astype(wide), y_ref).real) <= tol * (op.inf_norm() + 1.0) * n
    assert abs(op.inf_norm() - np.max(np.sum(np.abs(a.astype(wide)), axis=1))) <= 1e-12 * n
    op.close()


@pytest.mark.parametrize("name", ["simple_matrix", "hermitian_matrix", "multiple_eigenpairs", "single_element"])
def test_dense_known_answers(ctx, name):
    """T1:128-161, T1:375-409, T1:442-488, T1:411-440 with the matrix handed over as a dense block."""
    case = cases.eigen_cases()[name]
    rp, ci, va = case["csr"]
    n = rp.shape[0] - 1
    a = va.reshape(n, n)
    op = L.DenseOperator(ctx, a)
    eng = L.LambdaLanczos(op, n, case["find_maximum"], case["num_eigs"])
    if case["eps"] is not None:
        eng.eps = case["eps"]
    eng.eigenvalue_offset = case["offset"]
    eng.init_vector = lambda v, *_: v.__setitem__(slice(None), G.start_vector(n, 1, va.dtype))
    vals, vecs = eng.run()
    for r, want in enumerate(case["values"]):
        assert abs(vals[r] - want) <= max(abs(want) * eng.eps, 1e-8 if eng.eps > 1e-8 else 0.0)
        assert 1 - overlap(vecs[r], case["vectors"][r]) <= max(100 * eng.eps, 1e-12)
    op.close()


# ------------------------------------------------------------------ the 2-D tiled SpMV kernel inside the loops
@pytest.mark.parametrize("dtype", [np.float64, np.float32], ids=["d", "s"])
def test_lanczos_on_the_tiled_kernel_matches_the_oracle(ctx, oracle, dtype):
    """A banded matrix (n = 2e5, columns within +-300: eligible for the tiled kernel on its own, no test hook), the kernel
    chosen through ll_csr_options.kernel: a 60-iteration LambdaLanczos window in the default geometry — 1.6 MB vectors, i.e. the
    two-iterations-per-sweep form, whose operator inputs are UNNORMALISED vectors the kernel scales while it stages its x tiles —
    against the oracle (alpha / beta, Ritz pair), plus ll_spmv on an unaligned x pointer (the scalar staging path)."""
    n = 200_000
    rp, ci, va = G.randsym(n, band=300)
    va = va.astype(dtype)
    csr = (rp, ci, va)
    op = L.CsrOperator(ctx, *csr, kernel=capi.SPMV_TILED)
    assert op.selected_spmv() == capi.SPMV_TILED and op.accuracy() == capi.ACCURACY_NORMWISE
    init = G.start_vector(n, 1).astype(dtype)
    eng = L.LambdaLanczos(op, n, True, 1)
    eng.max_iteration = 60
    eng.init_vector = lambda v, *_: np.copyto(v, init)
    vals, vecs = eng.run()
    wide = (rp, ci, va.astype(np.float64))
    ora = oracle.lanczos(wide, init.astype(np.float64), True, max_iteration=60)
    single = dtype == np.float32
    tol = 2e-4 if single else 1e-10
    m = 12 if single else 60
    assert eng.getIterationCounts() == [60]
    assert np.max(np.abs(eng.last_alpha[:m] - ora["alpha"][:m])) <= tol * 30
    assert np.max(np.abs(eng.last_beta[:m - 1] - ora["beta"][:m - 1])) <= tol * 30
    assert abs(vals[0] - ora["eigenvalues"][0]) <= (2e-3 if single else 1e-10) * abs(vals[0])
    assert 1 - overlap(vecs[0].astype(np.float64), ora["eigenvectors"][0]) <= (1e-4 if single else 1e-8)
    if not single:
        assert eng.last_stats["pair_iterations"] >= 56
    # unaligned input pointer: x one element into a buffer
    x = G.start_vector(n + 1, 5).astype(dtype)
    xd, yd = ctx.to_device(x), ctx.empty(n, dtype)
    import types
    dot = L.spmv(op, types.SimpleNamespace(ptr=xd.at(1).value), yd, offset=-0.75, want_dot=True)
    y_ref = oracle.spmv(wide, x[1:].astype(np.float64)) - 0.75 * x[1:].astype(np.float64)
    assert np.max(np.abs(yd.get().astype(np.float64) - y_ref)) <= (3e-6 if single else 1e-13) * 40
    assert abs(dot - float(x[1:].astype(np.float64) @ y_ref)) <= (1e-5 if single else 1e-11) * n
    op.close()


def test_exponentiator_on_the_tiled_kernel_matches_the_oracle(ctx, oracle, llenv):
    """Complex Hermitian torus 200 x 200 (config 5's matrix in small) through the tiled kernel (forced for this small matrix:
    LL_TL_FORCE): exp(-iH) v against the oracle's Exponentiator::run."""
    llenv.setenv("LL_TL_FORCE", "1")
    N = 200
    csr = G.torus(N)
    op = L.CsrOperator(ctx, *csr, kernel=capi.SPMV_TILED)
    assert op.selected_spmv() == capi.SPMV_TILED
    inp = G.start_vector(N * N, 1, np.complex128)
    out, it = L.Exponentiator(op, N * N).run(-1j, inp)
    o_out, o_it, _ = oracle.expo(csr, -1j, inp)
    assert abs(it - o_it) <= 1
    assert np.max(np.abs(out - o_out)) <= 1e-10 * np.linalg.norm(inp)
    op.close()


def test_a_matrix_without_column_locality_is_not_eligible_for_the_tiled_kernel(ctx):
    """The random matrix of config 3 in small: its row blocks touch every column tile — the creation-time timing must not even
    consider the tiled kernel, and asking for it by name is an error, not a silent fallback."""
    csr = G.randsym(400_000)
    op = L.CsrOperator(ctx, *csr)
    assert op.selected_spmv() in (capi.SPMV_CSR_STREAM, capi.SPMV_PB) and op.autotune_ms_of(capi.SPMV_TILED) < 0
    op.close()
    with pytest.raises(L.LanczosHipError):
        L.CsrOperator(ctx, *csr, kernel=capi.SPMV_TILED)
