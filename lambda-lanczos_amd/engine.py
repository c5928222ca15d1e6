"""Host-side mirror of the reference's public interface for the Krylov hot path, on top of the C ABI.

``LambdaLanczos`` / ``Exponentiator`` keep the reference's names, constructor shapes, public fields, defaults and
return conventions (include/lambda_lanczos/lambda_lanczos.hpp:109-415, exponentiator.hpp:24-211) so that the parity
tests read like the reference's own tests; the C++ twin of this file is include/lambda_lanczos_hip/*.hpp.
All numerics run in liblanczos_hip.so on the GPU — there is no CPU path in this package.
"""
import ctypes as C
import weakref

import numpy as np

from . import _capi as capi
from ._capi import check, lib, ptr

_EPS = float(np.finfo(np.float64).eps)


def _suffix(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float64:
        return "d"
    if dtype == np.complex128:
        return "z"
    if dtype == np.float32:
        return "s"
    if dtype == np.complex64:
        return "c"
    raise TypeError("supported scalar types: float32/64, complex64/128 (got %s)" % dtype)


def _is_complex(dtype):
    return np.dtype(dtype).kind == "c"


def _real_eps(dtype):
    """Machine epsilon of real_t<T> (the reference scales its default tolerances with it, LL:150, EX:58)."""
    return float(np.finfo(np.dtype(dtype)).eps)


class DeviceArray:
    """A device allocation owned by a Context (freed with it or by .free())."""

    def __init__(self, ctx, shape, dtype):
        self.ctx, self.shape, self.dtype = ctx, tuple(np.atleast_1d(shape)), np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        p = C.c_void_p()
        check(lib().ll_malloc(ctx.handle, self.nbytes, C.byref(p)))
        self.ptr = p.value

    def at(self, elem_offset):
        return C.c_void_p(self.ptr + int(elem_offset) * self.dtype.itemsize)

    def set(self, host):
        host = np.ascontiguousarray(host, dtype=self.dtype)
        assert host.nbytes <= self.nbytes
        check(lib().ll_memcpy_h2d(self.ctx.handle, self.ptr, ptr(host), host.nbytes))
        return self

    def get(self):
        out = np.empty(self.shape, dtype=self.dtype)
        check(lib().ll_memcpy_d2h(self.ctx.handle, ptr(out), self.ptr, self.nbytes))
        return out

    def free(self):
        if self.ptr:
            check(lib().ll_free(self.ctx.handle, self.ptr))
            self.ptr = None


_LIVE_CONTEXTS = weakref.WeakSet()


def live_contexts():
    """The Context objects that have not been closed (the test suite reloads their LL_* switches after changing one)."""
    return [c for c in list(_LIVE_CONTEXTS) if c.handle]


# callables(context) run at the end of Context.__init__ (the test harness applies its per-context tuning through them)
CONTEXT_CREATED_HOOKS = []


class Context:
    """Device + HIP stream + workspace (+ RCCL communicator): ll_context."""

    def __init__(self, device=0, stream=None):
        h = C.c_void_p()
        if stream is None:
            check(lib().ll_ctx_create(int(device), C.byref(h)))
        else:
            check(lib().ll_ctx_create_on_stream(int(device), C.c_void_p(int(stream)), C.byref(h)))
        self.handle = h
        self.device = int(device)
        self.rank, self.n_ranks = 0, 1
        _LIVE_CONTEXTS.add(self)
        for hook in list(CONTEXT_CREATED_HOOKS):
            hook(self)

    # ---- multi-GPU
    @staticmethod
    def unique_id():
        buf = (C.c_char * capi.UNIQUE_ID_BYTES)()
        check(lib().ll_comm_unique_id(buf))
        return bytes(buf)

    def init_comm(self, unique_id, rank, n_ranks):
        buf = (C.c_char * capi.UNIQUE_ID_BYTES).from_buffer_copy(unique_id)
        check(lib().ll_comm_init(self.handle, buf, int(rank), int(n_ranks)))
        self.rank, self.n_ranks = int(rank), int(n_ranks)

    def ranks_seen(self):
        """Rank tags that arrived in the communicator self-check of init_comm (== n_ranks when healthy; 1 without one)."""
        out = C.c_int()
        check(lib().ll_comm_ranks_seen(self.handle, C.byref(out)))
        return out.value

    def transport(self):
        """Which transport answers the collectives: "rccl", "plugin:<path>", "attached" or "none" (ll_comm_transport)."""
        buf = C.create_string_buffer(512)
        check(lib().ll_comm_transport(self.handle, buf, len(buf)))
        return buf.value.decode()

    def partition(self, n):
        return partition(n, self.n_ranks, self.rank)

    # ---- memory
    def empty(self, shape, dtype=np.float64):
        return DeviceArray(self, shape, dtype)

    def to_device(self, host):
        host = np.ascontiguousarray(host)
        return DeviceArray(self, host.shape, host.dtype).set(host)

    def synchronize(self):
        check(lib().ll_ctx_synchronize(self.handle))

    def stream(self):
        p = C.c_void_p()
        check(lib().ll_ctx_stream(self.handle, C.byref(p)))
        return p.value

    def timer_start(self):
        check(lib().ll_timer_start(self.handle))

    def timer_stop(self):
        """Milliseconds of device time on this context's stream since timer_start()."""
        ms = C.c_double()
        check(lib().ll_timer_stop(self.handle, C.byref(ms)))
        return ms.value

    def bandwidth_probe(self, nbytes=2 << 30):
        """(read-only GB/s, copy GB/s) of two plain streaming kernels over nbytes, measured now (ll_bandwidth_probe)."""
        r, c = C.c_double(), C.c_double()
        check(lib().ll_bandwidth_probe(self.handle, int(nbytes), C.byref(r), C.byref(c)))
        return r.value, c.value

    def release_cache(self):
        check(lib().ll_ctx_release_cache(self.handle))

    def reload_env(self):
        """Read the LL_* environment switches again (they are read once, when the context is created)."""
        check(lib().ll_ctx_reload_env(self.handle))

    def set_tuning(self, key, value):
        """ll_ctx_set_tuning (UNSTABLE; tests and probes): one tuning field of this context by key, on top of the environment;
        value None removes the setting again."""
        check(lib().ll_ctx_set_tuning(self.handle, str(key).encode(), None if value is None else str(value).encode()))

    def set_profiling(self, on):
        check(lib().ll_ctx_set_profiling(self.handle, 1 if on else 0))

    def close(self):
        if self.handle:
            check(lib().ll_ctx_destroy(self.handle))
            self.handle = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def partition(n, n_ranks, rank):
    """(row_begin, n_local) of `rank` in the 1-D contiguous row partition (ll_partition)."""
    b, c = C.c_int64(), C.c_int64()
    check(lib().ll_partition(int(n), int(n_ranks), int(rank), C.byref(b), C.byref(c)))
    return b.value, c.value


# ------------------------------------------------------------------ operators (the mv_mul plugin)
class _Operator:
    handle = None

    def info(self):
        n, nl, nnz = C.c_int64(), C.c_int64(), C.c_int64()
        check(lib().ll_op_info(self.handle, C.byref(n), C.byref(nl), C.byref(nnz)))
        return n.value, nl.value, nnz.value

    def close(self):
        if self.handle:
            check(lib().ll_op_destroy(self.handle))
            self.handle = None


class CsrOperator(_Operator):
    """Device-resident CSR matrix (rows [row_begin, row_begin+len(row_ptr)-1) of an n_cols x n_cols operator)."""

    def __init__(self, ctx, row_ptr, col, val, n_cols=None, row_begin=0, accuracy=None, kernel=None):
        """accuracy: None (the environment decides), capi.ACCURACY_NORMWISE or capi.ACCURACY_COMPONENTWISE — the accuracy class
        of y = A x (include/lanczos_hip.h, ll_csr_options); kernel: None (timed at creation), capi.SPMV_CSR_STREAM / SPMV_PB / SPMV_TILED
        (SPMV_TILED: an error for a matrix that is not eligible or has no entries, never a silent fallback)."""
        self.ctx = ctx
        row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int64)
        col = np.ascontiguousarray(col, dtype=np.int32)
        val = np.ascontiguousarray(val)
        self.dtype = val.dtype
        sfx = _suffix(val.dtype)
        n_rows = row_ptr.shape[0] - 1
        self.n = int(n_cols if n_cols is not None else n_rows)
        self.n_local, self.row_begin = n_rows, int(row_begin)
        h = C.c_void_p()
        if accuracy is None and kernel is None:
            fn = getattr(lib(), "ll_op_create_csr_" + sfx)
            check(fn(ctx.handle, n_rows, self.n, self.row_begin, ptr(row_ptr), ptr(col), ptr(val), C.byref(h)))
        else:
            opt = capi.CsrOptions()
            check(lib().ll_csr_options_default(C.byref(opt)))
            opt.accuracy = capi.ACCURACY_DEFAULT if accuracy is None else int(accuracy)
            opt.kernel = -1 if kernel is None else int(kernel)
            fn = getattr(lib(), "ll_op_create_csr_opt_" + sfx)
            check(fn(ctx.handle, n_rows, self.n, self.row_begin, ptr(row_ptr), ptr(col), ptr(val), C.byref(opt), C.byref(h)))
        self.handle = h
        self.nnz = int(row_ptr[-1])

    @classmethod
    def from_coo(cls, ctx, n, rows, cols, vals):
        """{row, col, value} triplets (sample2_sparse.cpp:14-47); conversion to CSR happens in the library."""
        self = cls.__new__(cls)
        rows = np.ascontiguousarray(rows, dtype=np.int32)
        cols = np.ascontiguousarray(cols, dtype=np.int32)
        vals = np.ascontiguousarray(vals)
        self.ctx, self.dtype, self.n, self.n_local, self.row_begin, self.nnz = ctx, vals.dtype, int(n), int(n), 0, len(vals)
        h = C.c_void_p()
        fn = getattr(lib(), "ll_op_create_coo_" + _suffix(vals.dtype))
        check(fn(ctx.handle, int(n), len(vals), ptr(rows), ptr(cols), ptr(vals), C.byref(h)))
        self.handle = h
        return self

    def inf_norm(self):
        """max_i sum_j |a_ij| of the local rows (a safe |eigenvalue_offset|)."""
        v = C.c_double()
        check(lib().ll_op_inf_norm(self.handle, C.byref(v)))
        return v.value

    def select_spmv(self, kind):
        """capi.SPMV_PB or capi.SPMV_CSR_STREAM (default: whichever timed faster at creation)."""
        check(lib().ll_op_select_spmv(self.handle, int(kind)))

    def set_accuracy(self, accuracy):
        """capi.ACCURACY_NORMWISE or capi.ACCURACY_COMPONENTWISE (ll_op_set_accuracy): same image, another summation kernel."""
        check(lib().ll_op_set_accuracy(self.handle, int(accuracy)))

    def accuracy(self):
        """The accuracy class of the kernel selected now (ll_op_accuracy)."""
        a = C.c_int()
        check(lib().ll_op_accuracy(self.handle, C.byref(a)))
        return a.value

    def autotune_ms(self):
        """(csr_stream_ms, pb_ms) measured when the operator was created (-1: not timed)."""
        a, b = C.c_double(), C.c_double()
        check(lib().ll_op_autotune_ms(self.handle, C.byref(a), C.byref(b)))
        return a.value, b.value

    def autotune_ms_of(self, kind):
        """Creation-time timing of one kernel (capi.SPMV_*), -1 when it was not a candidate."""
        a = C.c_double()
        check(lib().ll_op_autotune_ms_of(self.handle, int(kind), C.byref(a)))
        return a.value

    def tiled_layout(self):
        """(row blocks of the tiled image, those of them whose tiles all lie inside the rank's own columns) — (0, 0) without a tiled image."""
        a, b = C.c_int(), C.c_int()
        check(lib().ll_op_tiled_layout(self.handle, C.byref(a), C.byref(b)))
        return a.value, b.value

    def selected_spmv(self):
        k = C.c_int()
        check(lib().ll_op_selected_spmv(self.handle, C.byref(k)))
        return k.value


class DenseOperator(_Operator):
    """Dense row-major matrix (sample1_simple.cpp:22-28): rows [row_begin, row_begin + a.shape[0]) of an n x n matrix."""

    def __init__(self, ctx, a, row_begin=0):
        a = np.ascontiguousarray(a)
        if a.dtype not in (np.float32, np.float64, np.complex64, np.complex128):
            a = a.astype(np.float64)
        self.ctx, self.dtype = ctx, a.dtype
        self.n_local, self.n = int(a.shape[0]), int(a.shape[1])
        self.row_begin, self.nnz = int(row_begin), int(a.size)
        h = C.c_void_p()
        fn = getattr(lib(), "ll_op_create_dense_" + _suffix(a.dtype))
        check(fn(ctx.handle, self.n_local, self.n, self.row_begin, ptr(a), C.byref(h)))
        self.handle = h

    inf_norm = CsrOperator.inf_norm


class StencilOperator(_Operator):
    """Matrix-free lattice operator (sample3_dynamic.cpp:17-22, T1:265-273, T2:113-121):
    (A x)(r) = (diag + onsite[r]) x(r) + sum_d (hop[d] x(r+e_d) + conj(hop[d]) x(r-e_d)) on a row-major lattice
    `dims` (last index fastest), open or periodic per dimension; phase_grad (ndim x ndim, complex types) adds Peierls
    phases exp(i * phase_grad[d] . coords(lower site)) to the hops.  Sharded contexts pass their ll_partition range."""

    def __init__(self, ctx, dims, diag=0.0, hop=-1.0, periodic=False, onsite=None, dtype=np.float64, row_begin=0,
                 n_local=None, phase_grad=None):
        dims = [int(d) for d in np.atleast_1d(dims)]
        nd = len(dims)
        if not 1 <= nd <= 3:
            raise ValueError("the lattice operator supports 1, 2 or 3 dimensions")
        hop = np.broadcast_to(np.asarray(hop, dtype=np.complex128), (nd,))
        periodic = np.broadcast_to(np.asarray(periodic, dtype=bool), (nd,))
        d = capi.StencilDesc()
        d.ndim = nd
        for k in range(nd):
            d.dims[k], d.periodic[k] = dims[k], int(periodic[k])
            d.hop_re[k], d.hop_im[k] = float(hop[k].real), float(hop[k].imag)
        d.diag = float(diag)
        if phase_grad is not None:  # Peierls phases: bond r -> r+e_d carries hop[d] * exp(i * phase_grad[d] . coords(r))
            pg = np.asarray(phase_grad, dtype=np.float64).reshape(nd, nd)
            for k in range(nd):
                for e in range(nd):
                    d.phase_grad[k][e] = float(pg[k, e])
        self.ctx, self.dtype = ctx, np.dtype(dtype)
        self.n = int(np.prod(dims))
        self.n_local = self.n if n_local is None else int(n_local)
        self.row_begin, self.nnz = int(row_begin), 0
        os_ = None if onsite is None else np.ascontiguousarray(onsite, dtype=np.float64)
        if os_ is not None and os_.shape[0] != self.n_local:
            raise ValueError("onsite must hold n_local values")
        h = C.c_void_p()
        fn = getattr(lib(), "ll_op_create_stencil_" + _suffix(self.dtype))
        check(fn(ctx.handle, C.byref(d), self.row_begin, self.n_local, ptr(os_), C.byref(h)))
        self.handle = h

    inf_norm = CsrOperator.inf_norm


class HostOperator(_Operator):
    """Unmodified user code: mv_mul(in, out) on numpy arrays, `out` zero-filled on entry (LL:120-126)."""

    def __init__(self, ctx, mv_mul, n, dtype=np.float64):
        self.ctx, self.n, self.n_local, self.row_begin = ctx, int(n), int(n), 0
        self.dtype = np.dtype(dtype)
        self.nnz = 0
        self.calls = 0
        sfx = _suffix(dtype)
        dt = self.dtype

        def tramp(in_p, out_p, nn, _user):
            try:
                a = np.frombuffer((C.c_char * (nn * dt.itemsize)).from_address(in_p), dtype=dt)
                b = np.frombuffer((C.c_char * (nn * dt.itemsize)).from_address(out_p), dtype=dt)
                self.calls += 1
                mv_mul(a, b)
                return 0
            except Exception:  # noqa: BLE001 - reported through the C ABI as LL_ERR_CALLBACK
                import traceback

                traceback.print_exc()
                return 1

        self._cb = capi.HOST_MV_FN(tramp)  # keep alive
        h = C.c_void_p()
        check(getattr(lib(), "ll_op_create_host_" + sfx)(ctx.handle, self.n, self._cb, None, C.byref(h)))
        self.handle = h


# ------------------------------------------------------------------ primitives (one call per kernel family)
def spmv(op, x_dev, y_dev, offset=0.0, want_dot=False):
    d = C.c_double()
    fn = getattr(lib(), "ll_spmv_" + _suffix(op.dtype))
    check(fn(op.ctx.handle, op.handle, x_dev.ptr, y_dev.ptr, float(offset), C.byref(d) if want_dot else None))
    return d.value if want_dot else None


def dot(ctx, a_dev, b_dev, n=None):
    sfx = _suffix(a_dev.dtype)
    n = int(n if n is not None else a_dev.shape[-1])
    out = (C.c_double * 2)()
    check(getattr(lib(), "ll_dot_" + sfx)(ctx.handle, n, a_dev.ptr, b_dev.ptr, out))
    return out[0] if sfx in "ds" else complex(out[0], out[1])


def nrm2(ctx, v_dev, n=None):
    n = int(n if n is not None else v_dev.shape[-1])
    out = C.c_double()
    check(getattr(lib(), "ll_nrm2_" + _suffix(v_dev.dtype))(ctx.handle, n, v_dev.ptr, C.byref(out)))
    return out.value


def scal(ctx, a, v_dev, n=None):
    n = int(n if n is not None else v_dev.shape[-1])
    check(getattr(lib(), "ll_scal_" + _suffix(v_dev.dtype))(ctx.handle, n, float(a), v_dev.ptr))


def normalize(ctx, v_dev, n=None):
    n = int(n if n is not None else v_dev.shape[-1])
    out = C.c_double()
    check(getattr(lib(), "ll_normalize_" + _suffix(v_dev.dtype))(ctx.handle, n, v_dev.ptr, C.byref(out)))
    return out.value


def three_term(ctx, w_dev, u_prev_dev, u_cur_dev, beta, alpha, n=None):
    n = int(n if n is not None else w_dev.shape[-1])
    fn = getattr(lib(), "ll_three_term_" + _suffix(w_dev.dtype))
    check(fn(ctx.handle, n, w_dev.ptr, None if u_prev_dev is None else u_prev_dev.ptr, u_cur_dev.ptr, float(beta),
             float(alpha)))


def orth_block(ctx, basis_dev, nb, ld, w_dev, n, mode=capi.ORTH_CGS_DGKS, want_h=False):
    sfx = _suffix(w_dev.dtype)
    norm = C.c_double()
    cplx = sfx in "zc"
    h = np.zeros(max(nb, 1) * (2 if cplx else 1), dtype=np.float64) if want_h else None
    fn = getattr(lib(), "ll_orth_block_" + sfx)
    check(fn(ctx.handle, int(n), int(nb), None if basis_dev is None else basis_dev.ptr, int(ld), w_dev.ptr, int(mode),
             C.byref(norm), ptr(h)))
    if want_h:
        hh = h[: nb * (2 if cplx else 1)]
        return norm.value, (hh.view(np.complex128) if cplx else hh)
    return norm.value


def gemv_basis(ctx, basis_dev, m, ld, coeff, out_dev, ld_out, n):
    sfx = _suffix(out_dev.dtype)
    if sfx in "sc":   # the float entry points take their coefficients as doubles, like every scalar
        coeff = np.ascontiguousarray(coeff, dtype=np.complex128 if sfx == "c" else np.float64)
    else:
        coeff = np.ascontiguousarray(coeff, dtype=out_dev.dtype)
    nout = coeff.shape[0] if coeff.ndim == 2 else 1
    fn = getattr(lib(), "ll_gemv_basis_" + sfx)
    check(fn(ctx.handle, int(n), int(m), basis_dev.ptr, int(ld), int(nout), ptr(coeff), out_dev.ptr, int(ld_out)))


def tridiag_eig(alpha, beta, want_vectors=True):
    """Host tridiagonal eigen-solver of the library (a11): ascending eigenvalues, rows of q = eigenvectors."""
    alpha = np.ascontiguousarray(alpha, dtype=np.float64)
    m = alpha.shape[0]
    beta = np.ascontiguousarray(np.concatenate([np.asarray(beta, dtype=np.float64), np.zeros(1)]))
    ev = np.empty(m)
    q = np.empty((m, m)) if want_vectors else None
    unc = C.c_int64()
    check(lib().ll_tridiag_eig(m, ptr(alpha), ptr(beta), ptr(ev), ptr(q), C.byref(unc)))
    return (ev, q, unc.value) if want_vectors else (ev, unc.value)


def tridiag_eigvecs(alpha, beta, lambdas):
    """Eigenvectors (rows) of T(alpha, beta) for the given eigenvalues by inverse iteration."""
    alpha = np.ascontiguousarray(alpha, dtype=np.float64)
    beta = np.ascontiguousarray(np.concatenate([np.asarray(beta, dtype=np.float64), np.zeros(1)]))
    lam = np.ascontiguousarray(np.atleast_1d(lambdas), dtype=np.float64)
    out = np.empty((lam.shape[0], alpha.shape[0]))
    check(lib().ll_tridiag_eigvecs(alpha.shape[0], ptr(alpha), ptr(beta), lam.shape[0], ptr(lam), ptr(out)))
    return out


def tridiag_bisect(alpha, beta, k):
    alpha = np.ascontiguousarray(alpha, dtype=np.float64)
    beta = np.ascontiguousarray(np.concatenate([np.asarray(beta, dtype=np.float64), np.zeros(1)]))
    out = C.c_double()
    check(lib().ll_tridiag_bisect(alpha.shape[0], ptr(alpha), ptr(beta), int(k), C.byref(out)))
    return out.value


# ------------------------------------------------------------------ the engines
_default_context = None


def default_context():
    global _default_context
    if _default_context is None:
        _default_context = Context(0)
    return _default_context


def _as_operator(mv_mul, n, dtype, ctx):
    if isinstance(mv_mul, _Operator):
        return mv_mul, False
    if callable(mv_mul):
        return HostOperator(ctx, mv_mul, n, dtype), True
    raise TypeError("mv_mul must be a CsrOperator/HostOperator or a callable mv_mul(in, out)")


class LambdaLanczos:
    """lambda_lanczos::LambdaLanczos<T> (LL:109-415) with device-resident Krylov vectors.

    ``mv_mul`` is a device operator (``CsrOperator``) or — for unmodified user code — a callable
    ``mv_mul(in_array, out_array)`` with the reference's contract (out zero-filled, accumulate or overwrite).
    Public fields and defaults are the reference's (LL:126-181)."""

    def __init__(self, mv_mul, matrix_size, find_maximum, num_eigs, dtype=None, context=None):
        self.context = context or (mv_mul.ctx if isinstance(mv_mul, _Operator) else default_context())
        self.dtype = np.dtype(dtype if dtype is not None else getattr(mv_mul, "dtype", np.float64))
        self.mv_mul = mv_mul                                     # LL:126
        self.init_vector = None                                  # LL:133 (None = random default, LL:70-104)
        self.matrix_size = int(matrix_size)                      # LL:136
        self.max_iteration = int(matrix_size)                    # LL:138,206
        self.eps = _real_eps(self.dtype) * 1e3                   # LL:150 (epsilon of real_t<T>)
        self.find_maximum = bool(find_maximum)                   # LL:153
        self.num_eigs = int(num_eigs)                            # LL:156
        self.eigenvalue_offset = 0.0                             # LL:165
        self.num_eigs_per_iteration = 5                          # LL:173
        self.initial_vector_size = 200                           # LL:181
        # additions (0 = reference-faithful)
        self.tridiag_mode = capi.TRIDIAG_AUTO  # decision- and value-identical to the reference's QR, O(k) per iteration
        self.orth_mode = capi.ORTH_CGS_DGKS
        # device-resident I/O (additions): init_vector may be a DeviceArray (start vector already in HBM), and a
        # DeviceArray of shape (num_eigs, n_local) here receives the eigenvectors instead of a host array
        self.eigenvectors_out = None
        self._iter_counts = []
        self.last_stats = None
        self.last_alpha = self.last_beta = None

    def _params(self, num_eigs):
        p = capi.LanczosParams()
        check(lib().ll_lanczos_params_default(C.byref(p), self.matrix_size, int(self.find_maximum), int(num_eigs)))
        p.max_iteration = int(self.max_iteration)
        p.eps = float(self.eps)
        p.eigenvalue_offset = float(self.eigenvalue_offset)
        p.num_eigs_per_iteration = int(self.num_eigs_per_iteration)
        p.initial_vector_size = int(self.initial_vector_size)
        p.tridiag_mode = int(self.tridiag_mode)
        p.orth_mode = int(self.orth_mode)
        return p

    def run(self, num_eigs=None):
        """Returns (eigenvalues, eigenvectors): eigenvectors[k] is the k-th eigenvector (LL:330-386).
        For sharded contexts each rank receives its row shard of every eigenvector."""
        k = int(self.num_eigs if num_eigs is None else num_eigs)
        vals, vecs, _ = self._drive(k, None, None)
        return vals, vecs

    def run_iteration(self, nroot, orthogonalize_to=None):
        """LambdaLanczos<T>::run_iteration(eigvalues, eigvecs, nroot, orthogonalizeTo) (LL:216-322): ONE pass tracking
        nroot Ritz pairs, every Lanczos vector orthogonalised against the rows of orthogonalize_to first.
        Returns (eigenvalues, eigenvectors, iteration_count); no restart loop, no EigenPairManager filtering."""
        return self._drive(int(nroot), int(nroot), orthogonalize_to)

    def _drive(self, k, nroot, orth):
        op, owned = _as_operator(self.mv_mul, self.matrix_size, self.dtype, self.context)
        sfx = _suffix(self.dtype)
        p = self._params(1 if nroot is not None else k)
        keep = None
        if isinstance(self.init_vector, DeviceArray):  # start vector already in HBM
            assert self.init_vector.dtype == self.dtype and self.init_vector.nbytes >= op.n_local * self.dtype.itemsize
            p.init_vector_dev = self.init_vector.ptr
        elif self.init_vector is not None:
            dt, user_fn = self.dtype, self.init_vector

            def tramp(vec_p, n_local, row_begin, _user):
                v = np.frombuffer((C.c_char * (n_local * dt.itemsize)).from_address(vec_p), dtype=dt)
                try:
                    user_fn(v, row_begin)
                except TypeError:
                    user_fn(v)  # reference signature init_vector(vec) (LL:133)

            keep = capi.INIT_FN(tramp)
            p.init_vector = keep
        n_local = op.n_local
        vals = np.zeros(k, dtype=np.float64)
        dev_out = self.eigenvectors_out
        if dev_out is not None:  # Ritz vectors stay in HBM: rows of the caller's DeviceArray
            assert dev_out.dtype == self.dtype and dev_out.nbytes >= k * n_local * self.dtype.itemsize
            vecs, vecs_p = dev_out, C.c_void_p(dev_out.ptr)
        else:
            vecs = np.empty((k, n_local), dtype=self.dtype)
            vecs_p = ptr(vecs)
        n_found = C.c_int64()
        cap = 4 * k + 64
        counts = np.zeros(cap, dtype=np.int64)
        trace_cap = int(min(self.max_iteration, 1 << 24))
        alpha = np.zeros(trace_cap)
        beta = np.zeros(trace_cap)
        stats = capi.RunStats()
        itern = C.c_int64()
        try:
            if nroot is None:
                fn = getattr(lib(), "ll_lanczos_run_" + sfx)
                check(fn(self.context.handle, op.handle, C.byref(p), ptr(vals), vecs_p, C.byref(n_found), ptr(counts),
                         cap, ptr(alpha), ptr(beta), C.byref(stats)))
            else:
                lock = None
                n_orth = 0
                lock_p = None
                if isinstance(orth, DeviceArray):  # orthogonalizeTo vectors already in HBM: rows of a (n_orth, n_local) array
                    assert orth.dtype == self.dtype and orth.shape[-1] == n_local
                    n_orth = int(np.prod(orth.shape[:-1])) if len(orth.shape) > 1 else 1
                    lock_p = C.c_void_p(orth.ptr)
                elif orth is not None and len(orth):
                    lock = np.ascontiguousarray(orth, dtype=self.dtype).reshape(-1, n_local)
                    n_orth = lock.shape[0]
                    lock_p = ptr(lock)
                fn = getattr(lib(), "ll_lanczos_run_iteration_" + sfx)
                check(fn(self.context.handle, op.handle, C.byref(p), nroot, n_orth, lock_p, ptr(vals), vecs_p,
                         C.byref(n_found), C.byref(itern), ptr(alpha), ptr(beta), C.byref(stats)))
                counts[0] = itern.value
        finally:
            if owned:
                op.close()
        del keep
        nf = n_found.value
        self._iter_counts = [int(c) for c in counts[: min(stats.n_passes, cap)]]
        self.last_stats = stats.as_dict()
        self.last_alpha, self.last_beta = alpha[: stats.last_alpha_len].copy(), beta[: stats.last_alpha_len].copy()
        return vals[:nf], (vecs if dev_out is not None else vecs[:nf]), int(itern.value)

    def run_single(self):
        """run(eigenvalue, eigenvector): one pair regardless of num_eigs (LL:394-407)."""
        vals, vecs = self.run(num_eigs=1)
        return vals[0], vecs[0]

    def getIterationCounts(self):  # noqa: N802 - reference name (LL:412-414)
        return list(self._iter_counts)


class Exponentiator:
    """lambda_lanczos::Exponentiator<T> (EX:24-211): output = exp(a*A) input by Krylov projection."""

    def __init__(self, mv_mul, matrix_size, dtype=None, context=None):
        self.context = context or (mv_mul.ctx if isinstance(mv_mul, _Operator) else default_context())
        self.dtype = np.dtype(dtype if dtype is not None else getattr(mv_mul, "dtype", np.float64))
        self.mv_mul = mv_mul                                     # EX:41
        self.matrix_size = int(matrix_size)                      # EX:44
        self.max_iteration = int(matrix_size)                    # EX:46,81
        self.eps = _real_eps(self.dtype) * 1e2                   # EX:58
        self.full_orthogonalize = False                          # EX:63
        self.initial_vector_size = 200                           # EX:71
        self.orth_mode = capi.ORTH_CGS_DGKS
        self.last_stats = None

    def _params(self):
        p = capi.ExpoParams()
        check(lib().ll_expo_params_default(C.byref(p), self.matrix_size))
        p.max_iteration = int(self.max_iteration)
        p.eps = float(self.eps)
        p.full_orthogonalize = int(bool(self.full_orthogonalize))
        p.orth_mode = int(self.orth_mode)
        p.initial_vector_size = int(self.initial_vector_size)
        return p

    def _call(self, name, a, input, want_stats, out=None):
        op, owned = _as_operator(self.mv_mul, self.matrix_size, self.dtype, self.context)
        sfx = _suffix(self.dtype)
        if isinstance(input, DeviceArray):  # psi stays in HBM: device input, device output (out= or a new DeviceArray)
            assert input.dtype == self.dtype and input.nbytes >= op.n_local * self.dtype.itemsize
            out = out if out is not None else DeviceArray(self.context, (op.n_local,), self.dtype)
            in_p, out_p = C.c_void_p(input.ptr), C.c_void_p(out.ptr)
        else:
            inp = np.ascontiguousarray(input, dtype=self.dtype)
            assert inp.shape[0] == op.n_local, "input size differs from the (local) matrix size (EX:88)"
            out = np.zeros_like(inp)
            in_p, out_p = ptr(inp), ptr(out)
        it = C.c_int64()
        p = self._params()
        stats = capi.RunStats()
        try:
            fn = getattr(lib(), name + sfx)
            args = [self.context.handle, op.handle, C.byref(p)]
            args += [float(a)] if sfx in "ds" else [float(np.real(a)), float(np.imag(a))]
            args += [in_p, out_p, C.byref(it)]
            if want_stats:
                args.append(C.byref(stats))
            check(fn(*args))
        finally:
            if owned:
                op.close()
        if want_stats:
            self.last_stats = stats.as_dict()
        return out, it.value

    def run(self, a, input, out=None):
        """Returns (output, iteration_count) (EX:87-173).  A DeviceArray input keeps the vector in HBM (output: `out`,
        which may be `input` itself, or a new DeviceArray)."""
        return self._call("ll_expo_run_", a, input, True, out)

    def taylor_run(self, a, input, out=None):
        """Returns (output, number_of_terms) (EX:175-210)."""
        return self._call("ll_expo_taylor_run_", a, input, False, out)
