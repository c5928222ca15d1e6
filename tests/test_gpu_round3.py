"""Round-3 GPU tests: the accuracy class of the fixed-point SpMV pinned on inputs where it differs from the floating-point
one, whole loops from a localized start vector against the REAL reference, BASELINE config 5 at n = 1e6 against the real
Exponentiator::run, taylor_run's a == 0 exit with device buffers, and a communicator whose self-check fails."""
import ctypes as C

import numpy as np
import pytest

import lambda_lanczos_amd as L
from lambda_lanczos_amd import _capi as capi
from lambda_lanczos_amd import generators as G
from util import inf_norm, overlap

pytestmark = pytest.mark.gpu
EPS = np.finfo(np.float64).eps


def fixed_init(v):
    return lambda out, *_: np.copyto(out, v)


# ------------------------------------------------------------------ accuracy class of LL_PB_PHASE2=fixed (lanczos_hip.h)
def _exact_rows(csr, x):
    """(A x)_i, sum_j |a_ij||x_j| and sum_j |a_ij| in extended precision (x87 long double: 64-bit mantissa); complex
    through separate real / imaginary parts."""
    rp, ci, va = csr
    ld = np.longdouble
    n = rp.shape[0] - 1
    rows = np.repeat(np.arange(n), np.diff(rp))

    def rowsum(v):
        out = np.zeros(n, dtype=ld)
        np.add.at(out, rows, v)
        return out

    xa = x[ci]
    if np.iscomplexobj(va) or np.iscomplexobj(x):
        ar, ai = np.real(va).astype(ld), np.imag(va).astype(ld)
        xr, xi = np.real(xa).astype(ld), np.imag(xa).astype(ld)
        y = (rowsum(ar * xr - ai * xi), rowsum(ar * xi + ai * xr))
        mag_a = np.abs(np.real(va)).astype(ld) + np.abs(np.imag(va)).astype(ld)
        mag_x = np.abs(xr) + np.abs(xi)
    else:
        y = (rowsum(va.astype(ld) * xa.astype(ld)), None)
        mag_a, mag_x = np.abs(va).astype(ld), np.abs(xa).astype(ld)
    return y, rowsum(mag_a * mag_x), rowsum(mag_a)


def _wide_inputs(kind, csr, dtype):
    rp, ci, va = csr
    n = rp.shape[0] - 1
    va = va.astype(dtype)
    if kind == "decades":            # x_j = 10^-(j mod 300): 300 decades of dynamic range inside every column block
        x = (10.0 ** (-(np.arange(n) % 300).astype(np.float64))).astype(dtype)
        if np.dtype(dtype) == np.complex128:
            x = x * np.exp(1j * np.arange(n))
    elif kind == "e0":               # a unit vector: the start vector of many users of this library
        x = np.zeros(n, dtype=dtype)
        x[0] = 1.0
    elif kind == "row_scales":       # D A with D_ii = 10^(+-140): the rows differ by 280 decades in scale
        rng = np.random.default_rng(3)
        x = rng.uniform(-1, 1, n).astype(dtype)
        if np.dtype(dtype) == np.complex128:
            x = x + 1j * rng.uniform(-1, 1, n)
        scale = np.where(np.arange(n) % 3 == 0, 1e140, np.where(np.arange(n) % 3 == 1, 1e-140, 1.0))
        va = va * np.repeat(scale, np.diff(rp))
    else:
        raise ValueError(kind)
    return (rp, ci, va), x


@pytest.mark.parametrize("block", [None, "37"], ids=["default_blocks", "small_blocks"])
@pytest.mark.parametrize("dtype", [np.float64, np.complex128], ids=["d", "z"])
@pytest.mark.parametrize("kind", ["decades", "e0", "row_scales"])
def test_fixed_point_spmv_meets_its_stated_normwise_bound(ctx, llenv, kind, dtype, block):
    """lanczos_hip.h states, for the default PB kernel (LL_ACCURACY_NORMWISE),
         |y_i - (A x)_i| <= eps sum_j |a_ij||x_j| + nnz_i 2^-60 (sum_j |a_ij|) max_k |x_k|,
    and the component-wise bound c nnz_i eps sum_j |a_ij||x_j| for LL_ACCURACY_COMPONENTWISE (CSR-stream, PB ordered).  On
    inputs with a huge dynamic range the two differ: both are asserted, each for the kernels it is stated for, and the
    `decades` case must actually separate them (some row of the fixed-point result violates the component-wise bound)."""
    csr, x = _wide_inputs(kind, G.randsym_np(5000), dtype)
    rp = csr[0]
    n = rp.shape[0] - 1
    nnz_i = np.diff(rp).astype(np.longdouble)
    (yr, yi), absrow, rowsum = _exact_rows(csr, x)
    xmax = np.longdouble(np.max(np.abs(np.real(x)) + np.abs(np.imag(x))))
    xd, yd = ctx.to_device(x), ctx.empty(n, dtype)

    def err_of(y):
        e = np.abs(np.real(y).astype(np.longdouble) - yr)
        if yi is not None:
            e = np.maximum(e, np.abs(np.imag(y).astype(np.longdouble) - yi))
        return e

    # The accuracy class is the CALLER's per-operator choice through the boundary (ll_csr_options.accuracy at creation,
    # ll_op_set_accuracy afterwards) — no environment variable selects the summation here.
    if block:
        llenv.setenv("LL_PB_BLOCK", block)   # (block geometry only: ragged blocks)
    got = {}
    for name, acc in (("fixed", capi.ACCURACY_NORMWISE), ("ordered", capi.ACCURACY_COMPONENTWISE)):
        op = L.CsrOperator(ctx, *csr, accuracy=acc, kernel=capi.SPMV_PB)
        assert op.selected_spmv() == capi.SPMV_PB and op.accuracy() == acc
        L.spmv(op, xd, yd)
        got[name] = yd.get()
        # the same operator moved to the other class and back: same image, the other summation kernel, the other class's bits
        other = capi.ACCURACY_COMPONENTWISE if acc == capi.ACCURACY_NORMWISE else capi.ACCURACY_NORMWISE
        op.set_accuracy(other)
        assert op.accuracy() == other
        L.spmv(op, xd, yd)
        got[name + "_switched"] = yd.get()
        op.set_accuracy(acc)
        L.spmv(op, xd, yd)
        assert np.array_equal(yd.get(), got[name])
        op.close()
    assert np.array_equal(got["fixed_switched"], got["ordered"]) and np.array_equal(got["ordered_switched"], got["fixed"])
    # the 2-D tiled kernel serves both classes out of ONE image (round 6): fixed-point sums = the PB kernel's bits, or the waves adding
    # in turn in floating point (component-wise); an operator moves between them like a PB operator
    llenv.setenv("LL_TL_FORCE", "1")
    op = L.CsrOperator(ctx, *csr, accuracy=capi.ACCURACY_COMPONENTWISE, kernel=capi.SPMV_TILED)
    assert op.selected_spmv() == capi.SPMV_TILED and op.accuracy() == capi.ACCURACY_COMPONENTWISE
    L.spmv(op, xd, yd)
    got["tiled_ordered"] = yd.get()
    L.spmv(op, xd, yd)
    assert np.array_equal(yd.get(), got["tiled_ordered"])          # a fixed order: the same bits on every launch
    op.set_accuracy(capi.ACCURACY_NORMWISE)
    assert op.accuracy() == capi.ACCURACY_NORMWISE
    L.spmv(op, xd, yd)
    assert np.array_equal(yd.get(), got["fixed"])                  # the integers of the PB kernel
    op.set_accuracy(capi.ACCURACY_COMPONENTWISE)
    L.spmv(op, xd, yd)
    assert np.array_equal(yd.get(), got["tiled_ordered"])
    op.close()
    llenv.delenv("LL_TL_FORCE")
    op = L.CsrOperator(ctx, *csr, kernel=capi.SPMV_CSR_STREAM)
    assert op.selected_spmv() == capi.SPMV_CSR_STREAM and op.accuracy() == capi.ACCURACY_COMPONENTWISE
    L.spmv(op, xd, yd)
    got["csr"] = yd.get()
    op.set_accuracy(capi.ACCURACY_NORMWISE)   # CSR-stream has one (component-wise) form: accepted, nothing changes
    assert op.accuracy() == capi.ACCURACY_COMPONENTWISE
    op.close()
    # default options == the plain constructor: norm-wise where PB is selected
    op = L.CsrOperator(ctx, *csr, kernel=capi.SPMV_PB)
    assert op.accuracy() == capi.ACCURACY_NORMWISE
    L.spmv(op, xd, yd)
    assert np.array_equal(yd.get(), got["fixed"])
    op.close()
    assert all(np.all(np.isfinite(v)) for v in got.values())
    tiny = np.longdouble(1e-320)
    componentwise = 8 * EPS * (nnz_i + 2) * absrow + tiny
    normwise = 2 * EPS * absrow + nnz_i * np.longdouble(2.0) ** -60 * rowsum * xmax + tiny
    assert np.all(err_of(got["ordered"]) <= componentwise)
    assert np.all(err_of(got["tiled_ordered"]) <= componentwise)
    assert np.all(err_of(got["csr"]) <= componentwise)
    assert np.all(err_of(got["fixed"]) <= normwise), float(np.max(err_of(got["fixed"]) / normwise))
    if kind == "decades":   # the inputs separate the two classes: fixed point is norm-wise accurate only
        assert np.any(err_of(got["fixed"]) > componentwise)
    if kind == "row_scales":  # the grid is per ROW: rows of very different scale each keep full relative accuracy
        assert np.all(err_of(got["fixed"]) <= componentwise)


@pytest.mark.parametrize("dtype", [np.float32, np.complex64], ids=["s", "c"])
def test_single_precision_pb_rounds_each_product_once_to_the_storage_type(ctx, llenv, dtype):
    """float / complex<float> PB: the product a_ij x_j goes to the product buffer in the storage type (one rounding, what a
    float multiply gives), the sum is fixed point / double: |y_i - exact| <= (eps_f / 2) sum |a_ij||x_j| (1 + small) plus
    the final rounding of y_i to float."""
    base = G.randsym_np(5000) if np.dtype(dtype) == np.float32 else G.torus_np(40)
    rp, ci = base[0], base[1]
    va = base[2].astype(dtype)
    n = rp.shape[0] - 1
    rng = np.random.default_rng(8)
    x = rng.uniform(-1, 1, n).astype(dtype)
    if np.dtype(dtype) == np.complex64:
        x = (x + 1j * rng.uniform(-1, 1, n)).astype(dtype)
    wide = np.complex128 if np.dtype(dtype) == np.complex64 else np.float64
    (yr, yi), absrow, _ = _exact_rows((rp, ci, va.astype(wide)), x.astype(wide))
    llenv.setenv("LL_SPMV_KERNEL", "pb")
    llenv.setenv("LL_PB_BLOCK", "257")
    op = L.CsrOperator(ctx, rp, ci, va)
    xd, yd = ctx.to_device(x), ctx.empty(n, dtype)
    L.spmv(op, xd, yd)
    y = yd.get()
    op.close()
    eps_f = np.finfo(np.float32).eps
    err = np.abs(np.real(y).astype(np.longdouble) - yr)
    if yi is not None:
        err = np.maximum(err, np.abs(np.imag(y).astype(np.longdouble) - yi))
    # complex: each of the four real products and the two sums of a complex product round in float
    per_product = (2.0 if yi is not None else 0.5) * eps_f
    assert np.all(err <= (per_product + 0.5 * eps_f) * absrow * 1.001 + 1e-30)


# ------------------------------------------------------------------ whole loops from a localized start vector vs the REAL reference
@pytest.mark.parametrize("kernel", ["pb", "csr"])
def test_lanczos_from_a_unit_start_vector_matches_the_real_reference(ctx, reference, llenv, kernel):
    """Start vector e_0 (max|x| = 1 while most entries of the first Lanczos vectors are exactly 0 or tiny): the
    norm-wise accuracy of the fixed-point SpMV is all the recurrence needs — same iteration count, eigenvalue and
    eigenvector as LambdaLanczos::run of the real reference with the same start vector."""
    n = 20011
    csr = G.randsym_np(n)
    init = np.zeros(n)
    init[0] = 1.0
    llenv.setenv("LL_SPMV_KERNEL", kernel)
    op = L.CsrOperator(ctx, *csr)
    eng = L.LambdaLanczos(op, n, True, 1)
    eng.init_vector = fixed_init(init)
    vals, vecs = eng.run()
    ref = reference.lanczos(csr, init, True)
    assert abs(eng.getIterationCounts()[0] - ref["iter_counts"][0]) <= 2
    assert abs(vals[0] - ref["eigenvalues"][0]) <= 1e-10 * abs(vals[0])
    assert 1 - overlap(vecs[0], ref["eigenvectors"][0]) <= 1e-8
    op.close()


def test_exponentiator_from_a_unit_input_vector_matches_the_real_reference(ctx, reference, llenv):
    """exp(-iH) e_0 on the complex torus (a wave packet spreading from one site: entries from 1 down to exact zeros)."""
    N = 48
    n = N * N
    csr = G.torus_np(N)
    psi = np.zeros(n, dtype=np.complex128)
    psi[0] = 1.0
    llenv.setenv("LL_SPMV_KERNEL", "pb")
    llenv.setenv("LL_PB_BLOCK", "300")
    op = L.CsrOperator(ctx, *csr)
    assert op.selected_spmv() == capi.SPMV_PB
    out, it = L.Exponentiator(op, n).run(-1j, psi)
    r_out, r_it, _ = reference.expo(csr, -1j, psi)
    assert it == r_it
    assert np.max(np.abs(out - r_out)) <= 1e-12
    assert abs(np.linalg.norm(out) - 1.0) <= 1e-12
    op.close()


# ------------------------------------------------------------------ BASELINE config 5 at full size vs the REAL reference
@pytest.fixture(scope="module")
def c5():
    N = 1000
    return N * N, G.torus(N), G.start_vector_fast(N * N, 1, np.complex128)


@pytest.mark.parametrize("dt,iters", [(0.1, 7), (1.0, 15), (5.0, 38)])
def test_c5_full_size_output_matches_the_real_exponentiator(ctx, reference, c5, dt, iters):
    """Config 5 at n = 1e6 (complex Hermitian torus, exp(-i H dt) v): iteration counts equal the reference's (7 / 15 / 38,
    SURVEY 3.4), max |out - out_ref| <= 1e-10 |in|, 1 - overlap <= 10 eps — against Exponentiator::run of the REAL
    reference (EX:87-173) through oracle/_ref, about 3 s of host time for the three exponents."""
    n, csr, init = c5
    op = L.CsrOperator(ctx, *csr)
    out, it = L.Exponentiator(op, n).run(-1j * dt, init)
    r_out, r_it, _ = reference.expo(csr, -1j * dt, init)
    assert it == r_it == iters
    assert np.max(np.abs(out - r_out)) <= 1e-10 * np.linalg.norm(init)
    assert 1 - overlap(out, r_out) <= 10 * EPS
    assert abs(np.linalg.norm(out) / np.linalg.norm(init) - 1.0) <= 1e-12
    op.close()


# ------------------------------------------------------------------ taylor_run, a == 0 (EX:179-182) with device buffers
@pytest.mark.parametrize("dtype", [np.float64, np.complex128], ids=["d", "z"])
def test_taylor_run_zero_exponent_with_device_buffers(ctx, dtype):
    """The a == 0 early exit copies input to output: both may live in HBM (a host memcpy would fault) and may be the SAME
    buffer (nothing to copy)."""
    n = 4099
    csr = G.randsym_np(n)
    op = L.CsrOperator(ctx, csr[0], csr[1], csr[2].astype(dtype))
    x = G.start_vector(n, 2, dtype)
    ex = L.Exponentiator(op, n)
    out_h, terms = ex.taylor_run(0.0, x)                      # host in, host out
    assert terms == 1 and np.array_equal(out_h, x)
    xd = ctx.to_device(x)
    out_d, terms = ex.taylor_run(0.0, xd)                     # device in, new device out
    assert terms == 1 and np.array_equal(out_d.get(), x) and out_d.ptr != xd.ptr
    same, terms = ex.taylor_run(0.0, xd, out=xd)              # in place
    assert terms == 1 and same.ptr == xd.ptr and np.array_equal(xd.get(), x)
    # and the Krylov form agrees (a = 0: identity, itern = 2 in the reference; EX:154 stops on the first overlap test)
    out_k, _ = ex.run(0.0, xd)
    assert np.max(np.abs(out_k.get() - x)) <= 1e-14 * np.max(np.abs(x))
    op.close()


# ------------------------------------------------------------------ a communicator whose self-check fails is detached again
def test_failed_communicator_self_check_leaves_the_context_unsharded():
    """ll_comm_attach runs the rank self-check; a transport that delivers nothing fails it with LL_ERR_RCCL.  The context
    must then be what it was before — no communicator, rank 0 of 1 — so that it stays usable and the attach can be retried."""
    calls = {"gather": 0, "reduce": 0, "destroy": 0}
    GATHER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
    REDUCE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
    HALO = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_size_t,
                       C.c_void_p)
    DESTROY = C.CFUNCTYPE(None, C.c_void_p)

    class Transport(C.Structure):
        _fields_ = [("self", C.c_void_p), ("all_gather", GATHER), ("all_reduce_sum_f64", REDUCE), ("halo_exchange", HALO),
                    ("destroy", DESTROY)]

    def gather(_self, _send, _recv, _bytes, _stream):
        calls["gather"] += 1
        return 0          # claims success, moves nothing: the rank tags never arrive

    def reduce(_self, _buf, _count, _stream):
        calls["reduce"] += 1
        return 0

    def halo(*_a):
        return 0

    def destroy(_self):
        calls["destroy"] += 1

    t = Transport(None, GATHER(gather), REDUCE(reduce), HALO(halo), DESTROY(destroy))
    c = L.Context(0)
    rc = capi.lib().ll_comm_attach(c.handle, C.byref(t), 0, 2)
    assert rc == capi.LL_ERR_RCCL and b"self-check" in capi.lib().ll_last_error()
    assert calls["gather"] == 1 and calls["destroy"] == 1
    r, w = C.c_int(-1), C.c_int(-1)
    capi.check(capi.lib().ll_comm_rank(c.handle, C.byref(r), C.byref(w)))
    assert (r.value, w.value) == (0, 1)
    # the context works as an ordinary single-GPU context ...
    n = 3001
    csr = G.randsym_np(n)
    op = L.CsrOperator(c, *csr)
    eng = L.LambdaLanczos(op, n, True, 1)
    eng.max_iteration = 20
    vals, _ = eng.run()
    assert np.isfinite(vals[0])
    op.close()
    # ... and a second attach is not refused with "communicator already attached" (it fails its self-check again)
    rc = capi.lib().ll_comm_attach(c.handle, C.byref(t), 0, 2)
    assert rc == capi.LL_ERR_RCCL and b"self-check" in capi.lib().ll_last_error()
    c.close()


# ------------------------------------------------------------------ lagged (one-sweep) block Gram-Schmidt vs the two-sweep form
def _lagged_case(name):
    if name == "randsym":
        n = 30011
        return n, G.randsym_np(n), G.start_vector(n, 1), True, 0.0
    if name == "laplace":   # slowly converging: several hundred iterations
        m = 173
        return m * m, G.laplace2d_np(m), G.start_vector(m * m, 2), True, 0.0
    if name == "laplace_long":   # 1800 iterations (eps = 0 never stops): more columns than one 48 KB LDS block of partials
        m = 150                  # held in rounds 1-2, basis over ten slabs
        return m * m, G.laplace2d_np(m), G.start_vector(m * m, 4), False, 0.0
    N = 160                 # complex Hermitian torus, lowest eigenvalue through the reference's offset idiom
    return N * N, G.torus_np(N), G.start_vector(N * N, 3, np.complex128), False, -10.0


@pytest.mark.parametrize("dtype", [np.float64, np.complex128, np.float32, np.complex64], ids=["d", "z", "s", "c"])
def test_one_sweep_form_in_the_small_vector_geometry(ctx, llenv, dtype):
    """Vectors between 320 KiB and 1 MiB take the one-sweep form through lagged_small_kernel (four waves per 1 KiB strip
    split the basis); LL_TEST_LAGGED_MIN_BYTES=0 puts a 30 011-row problem there.  Against the two-sweep small-vector
    kernels (LL_FUSE_LAUNCHES=1): same iteration count, traces to 1e-11 ||A|| (float: the first dozen to float rounding)."""
    single = dtype in (np.float32, np.complex64)
    wide = np.complex128 if dtype in (np.complex128, np.complex64) else np.float64
    n = 30011
    csr = G.randsym_np(n)
    csr = (csr[0], csr[1], csr[2].astype(dtype))
    init = G.start_vector(n, 1, wide).astype(dtype)
    llenv.setenv("LL_TEST_LAGGED_MIN_BYTES", "0")
    op = L.CsrOperator(ctx, *csr)
    got = {}
    for fuse in ("1", "2"):
        llenv.setenv("LL_FUSE_LAUNCHES", fuse)
        eng = L.LambdaLanczos(op, n, True, 2)
        eng.init_vector = lambda v, *_: np.copyto(v, init)
        vals, vecs = eng.run()
        got[fuse] = (vals, [v.astype(wide) for v in vecs], eng.getIterationCounts(), eng.last_alpha, eng.last_beta, eng.last_stats)
    two, one = got["1"], got["2"]
    assert two[5]["lagged_iterations"] == 0 and one[5]["lagged_iterations"] >= one[2][0] - 2
    scale = 30.0
    if single:
        assert abs(one[2][0] - two[2][0]) <= 2
        assert np.max(np.abs(one[3][:12] - two[3][:12])) <= 2e-4 * scale and np.max(np.abs(one[0] - two[0])) <= 2e-3 * scale
    else:
        assert one[2] == two[2]
        m = min(len(one[3]), len(two[3]))
        assert np.max(np.abs(one[3][:m] - two[3][:m])) <= 1e-11 * scale and np.max(np.abs(one[4][:m] - two[4][:m])) <= 1e-11 * scale
        assert np.max(np.abs(one[0] - two[0])) <= 1e-11 * scale
        for i in range(2):
            assert 1 - overlap(one[1][i], two[1][i]) <= 1e-9
    op.close()


@pytest.mark.parametrize("name", ["randsym", "laplace", "torus", "laplace_long"])
def test_lagged_gram_schmidt_keeps_the_recurrence_of_the_two_sweep_form(ctx, llenv, name):
    """The one-sweep Gram-Schmidt forms (since round 5 the default in the streaming geometry is TWO iterations per sweep, the pair
    form; LL_PAIR_GS=0 — set by the suite's pair-off leg, tools/r05_gpu_batch.sh — gives one sweep per iteration) against the
    two-sweep kernels on the same operator and start vector, streaming geometry forced on both: same iteration count, alpha /
    beta traces equal to 1e-11 ||A|| over the whole run (an uncompensated lag loses them after ~40 iterations), same eigenpair,
    residual at the level the Ritz estimate promises.  LL_FUSE_LAUNCHES=1 is the two-sweep comparator.  (tests/test_gpu_pair.py
    compares all three forms with each other and with the oracle.)"""
    n, csr, init, find_max, offset = _lagged_case(name)
    llenv.setenv("LL_BLAS_SMALL_BYTES", "0")
    op = L.CsrOperator(ctx, *csr)
    got = {}
    for fuse in ("1", "2"):
        llenv.setenv("LL_FUSE_LAUNCHES", fuse)
        eng = L.LambdaLanczos(op, n, find_max, 1)
        eng.eigenvalue_offset = offset
        eng.init_vector = fixed_init(init)
        if name == "laplace_long":
            eng.eps = 0.0
            eng.max_iteration = 1800
        vals, vecs = eng.run()
        got[fuse] = (vals[0], vecs[0], eng.getIterationCounts(), eng.last_alpha, eng.last_beta, eng.last_stats)
    two, one = got["1"], got["2"]
    if name == "laplace_long":
        assert one[2] == [1800]
    assert two[5]["lagged_iterations"] == 0
    assert one[5]["lagged_iterations"] >= one[2][0] - 2 - 2 * one[5]["second_passes"]
    assert one[2] == two[2]
    scale = inf_norm(csr) + abs(offset)
    if name == "laplace_long":
        # Past the convergence of the extreme Ritz values the alpha / beta sequence of ANY Lanczos process is an
        # ill-conditioned function of its rounding errors (two correct implementations drift apart; the drift here starts
        # around k = 500): compare the traces before that, and afterwards what the traces are for: the spectrum of T
        # (both ends: converged eigenvalues of A, known analytically) and its size
        import scipy.linalg as sl
        assert np.max(np.abs(one[3][:300] - two[3][:300])) <= 1e-11 * scale
        assert np.max(np.abs(one[4][:300] - two[4][:300])) <= 1e-11 * scale
        m = int(round(np.sqrt(n)))
        c = 2.0 - 2.0 * np.cos(np.arange(1, m + 1) * np.pi / (m + 1))
        exact = np.sort((c[:, None] + c[None, :]).ravel())
        for tr in (one, two):
            ritz = sl.eigvalsh_tridiagonal(tr[3], tr[4][:-1])
            assert np.max(np.abs(ritz[:40] - exact[:40])) <= 1e-11 * scale
            assert np.max(np.abs(ritz[-40:] - exact[-40:])) <= 1e-11 * scale
            assert np.all(tr[4] > 1e-3) and np.all(np.isfinite(tr[3]))
    else:
        assert np.max(np.abs(one[3] - two[3])) <= 1e-11 * scale
        assert np.max(np.abs(one[4] - two[4])) <= 1e-11 * scale
    assert abs(one[0] - two[0]) <= 1e-12 * scale
    assert 1 - overlap(one[1], two[1]) <= 1e-10
    res = np.linalg.norm(_csr_matvec(csr, one[1]) - one[0] * one[1])
    res2 = np.linalg.norm(_csr_matvec(csr, two[1]) - two[0] * two[1])
    assert res <= 2.0 * res2 + 1e-10 * scale
    op.close()


def _csr_matvec(csr, x):
    import scipy.sparse as sp
    rp, ci, v = csr
    return sp.csr_matrix((v, ci, rp), shape=(len(rp) - 1, len(rp) - 1)) @ x


@pytest.mark.parametrize("name,num_eigs", [("randsym", 3), ("torus", 4)])
def test_lagged_gram_schmidt_in_restart_passes_with_locked_eigenvectors(ctx, oracle, llenv, name, num_eigs):
    """Several eigenpairs (LL:334-354): the passes after the first orthogonalise against the locked eigenvectors (LL:233,259),
    whose image under the operator is lambda_i z_i: the one-sweep form compensates those columns with the eigenvalue.
    Same pass structure, iteration counts, eigenvalues and eigenvectors as the two-sweep form and as the oracle."""
    n, csr, init, find_max, offset = _lagged_case(name)
    llenv.setenv("LL_BLAS_SMALL_BYTES", "0")
    op = L.CsrOperator(ctx, *csr)
    got = {}
    for fuse in ("1", "2"):
        llenv.setenv("LL_FUSE_LAUNCHES", fuse)
        eng = L.LambdaLanczos(op, n, find_max, num_eigs)
        eng.eigenvalue_offset = offset
        eng.init_vector = fixed_init(init)
        vals, vecs = eng.run()
        got[fuse] = (vals, vecs, eng.getIterationCounts(), eng.last_stats)
    two, one = got["1"], got["2"]
    ora = oracle.lanczos(csr, init, find_max, num_eigs=num_eigs, offset=offset)
    assert len(one[2]) >= 2                                    # at least one pass with locked vectors
    assert one[2] == two[2] == ora["iter_counts"]
    assert two[3]["lagged_iterations"] == 0
    # the first pass always; a later pass if the locked Ritz vectors' residuals allow the first-order treatment of their
    # columns (measured at the start of the pass: they do for the well separated top of the random matrix's spectrum,
    # not for the torus' clustered band edge)
    want = sum(one[2]) if name == "randsym" else one[2][0]
    assert one[3]["lagged_iterations"] >= want - 2 * len(one[2]) - 2 * one[3]["second_passes"]
    scale = inf_norm(csr) + abs(offset)
    assert np.max(np.abs(one[0] - two[0])) <= 1e-11 * scale
    assert np.max(np.abs(one[0] - ora["eigenvalues"])) <= 1e-10 * scale
    for i in range(num_eigs):
        assert 1 - overlap(one[1][i], two[1][i]) <= 1e-9
        assert 1 - overlap(one[1][i], ora["eigenvectors"][i]) <= 1e-8
        for j in range(i):
            assert abs(np.vdot(one[1][i], one[1][j])) <= 1e-9
    op.close()


@pytest.mark.parametrize("dtype", [np.float32, np.complex64], ids=["s", "c"])
def test_lagged_gram_schmidt_single_precision(ctx, llenv, dtype):
    """float / complex<float> storage through the one-sweep form (streaming geometry forced): against the two-sweep
    form at float tolerances."""
    wide = np.float64 if dtype == np.float32 else np.complex128
    csr = G.randsym_np(20011) if dtype == np.float32 else G.torus_np(120)
    csr = (csr[0], csr[1], np.ascontiguousarray(csr[2]).astype(dtype))
    n = csr[0].shape[0] - 1
    init = G.start_vector(n, 1, wide).astype(dtype)
    find_max, offset = (True, 0.0) if dtype == np.float32 else (False, -10.0)
    llenv.setenv("LL_BLAS_SMALL_BYTES", "0")
    op = L.CsrOperator(ctx, *csr)
    got = {}
    for fuse in ("1", "2"):
        llenv.setenv("LL_FUSE_LAUNCHES", fuse)
        eng = L.LambdaLanczos(op, n, find_max, 1)
        eng.eigenvalue_offset = offset
        eng.init_vector = lambda v, *_: np.copyto(v, init)
        vals, vecs = eng.run()
        got[fuse] = (vals[0], vecs[0].astype(wide), eng.getIterationCounts(), eng.last_alpha, eng.last_beta, eng.last_stats)
    two, one = got["1"], got["2"]
    f32 = float(np.finfo(np.float32).eps)
    assert two[5]["lagged_iterations"] == 0 and one[5]["lagged_iterations"] > 0
    assert abs(one[2][0] - two[2][0]) <= 2
    m = min(len(one[3]), len(two[3]), 12)       # float traces drift apart at the rate any float Lanczos pair does
    assert np.max(np.abs(one[3][:m] - two[3][:m])) <= 1e3 * f32 * 16
    assert abs(one[0] - two[0]) <= 20 * 1e3 * f32 * 16
    assert 1 - overlap(one[1], two[1]) <= 1e-3
    op.close()


@pytest.mark.parametrize("dtype", [np.float64, np.complex128, np.float32], ids=["d", "z", "s"])
def test_both_strip_geometries_of_the_one_sweep_kernel(ctx, llenv, dtype):
    """Vectors below 3.2 MB take 32 bytes per lane and vector (twice the workgroups), longer ones 64; LL_TEST_LAGGED_PIECES
    forces either on the same problem: different partial sums, same traces / eigenpair to rounding, same iteration count."""
    n = 150001
    csr = G.randsym_np(n)
    csr = (csr[0], csr[1], csr[2].astype(dtype))
    wide = np.complex128 if dtype == np.complex128 else np.float64
    init = G.start_vector(n, 1, wide).astype(dtype)
    llenv.setenv("LL_BLAS_SMALL_BYTES", "0")
    op = L.CsrOperator(ctx, *csr)
    got = []
    for pc in ("4", "2"):
        llenv.setenv("LL_TEST_LAGGED_PIECES", pc)
        eng = L.LambdaLanczos(op, n, True, 1)
        eng.init_vector = lambda v, *_: np.copyto(v, init)
        eng.max_iteration = 120
        vals, vecs = eng.run()
        assert eng.last_stats["lagged_iterations"] > 50
        got.append((vals[0], vecs[0].astype(wide), eng.getIterationCounts(), eng.last_alpha, eng.last_beta))
    a, b = got
    tol = 1e-12 if dtype != np.float32 else 2e-4
    assert a[2] == b[2]
    m = len(a[3]) if dtype != np.float32 else 12
    assert np.max(np.abs(a[3][:m] - b[3][:m])) <= tol * 30 and np.max(np.abs(a[4][:m] - b[4][:m])) <= tol * 30
    assert abs(a[0] - b[0]) <= tol * 30 and 1 - overlap(a[1], b[1]) <= (1e-10 if dtype != np.float32 else 1e-3)
    op.close()


@pytest.mark.parametrize("a", [-1j, -0.5], ids=["unitary", "real_exponent"])
def test_lagged_gram_schmidt_in_the_exponentiator_with_full_orthogonalisation(ctx, oracle, llenv, a):
    """Exponentiator::run with full_orthogonalize (EX:120-122) through the one-sweep form, against the oracle."""
    N = 150
    n = N * N
    csr = G.torus_np(N)
    psi = G.start_vector(n, 5, np.complex128)
    llenv.setenv("LL_BLAS_SMALL_BYTES", "0")
    op = L.CsrOperator(ctx, *csr)
    ex = L.Exponentiator(op, n)
    ex.full_orthogonalize = True
    if a == -0.5:
        ex.max_iteration = 40
    out, it = ex.run(a, psi)
    o_out, o_it, _ = oracle.expo(csr, a, psi, full_orthogonalize=True, **({"max_iteration": 40} if a == -0.5 else {}))
    assert ex.last_stats["lagged_iterations"] >= it - 2
    assert it == o_it
    assert np.max(np.abs(out - o_out)) <= 1e-11 * np.linalg.norm(psi) * max(1.0, np.linalg.norm(o_out) / np.linalg.norm(psi))
    op.close()
