"""ctypes binding of liblanczos_hip.so — the C ABI declared in include/lanczos_hip.h.

Nothing in here computes: it loads the HIP library and forwards.  If the library is missing the import of
``lambda_lanczos_amd`` still works (so that the build helper can be reached) but every call raises
``LanczosHipError`` — there is no CPU fallback."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LL_LIB_PATH") or os.path.join(_HERE, "lib", "liblanczos_hip.so")
GEN_PATH = os.path.join(_HERE, "lib", "libllgen.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "lanczos_hip.h")

LL_OK, LL_ERR_INVALID, LL_ERR_HIP, LL_ERR_RCCL, LL_ERR_ALLOC, LL_ERR_CALLBACK = range(6)
ORTH_CGS_DGKS, ORTH_CGS2, ORTH_MGS = 0, 1, 2
TRIDIAG_QR, TRIDIAG_BISECT, TRIDIAG_AUTO = 0, 1, 2
SPMV_CSR_STREAM, SPMV_PB, SPMV_TILED = 0, 1, 2
ACCURACY_DEFAULT, ACCURACY_NORMWISE, ACCURACY_COMPONENTWISE = 0, 1, 2
UNIQUE_ID_BYTES = 128


ABI_VERSION = (0, 5)  # LL_VERSION_MAJOR, LL_VERSION_MINOR of the include/lanczos_hip.h these mirrors were written against


class LanczosHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("liblanczos_hip error %d: %s" % (code, msg))
        self.code = code


i64, i32, f64, vp = C.c_int64, C.c_int32, C.c_double, C.c_void_p
INIT_FN = C.CFUNCTYPE(None, vp, i64, i64, vp)
HOST_MV_FN = C.CFUNCTYPE(C.c_int, vp, vp, i64, vp)
DEV_MV_FN = C.CFUNCTYPE(C.c_int, vp, vp, i64, vp, vp)


class LanczosParams(C.Structure):
    _fields_ = [
        ("matrix_size", i64),
        ("max_iteration", i64),
        ("eps", f64),
        ("find_maximum", i32),
        ("reserved0", i32),
        ("num_eigs", i64),
        ("eigenvalue_offset", f64),
        ("num_eigs_per_iteration", i64),
        ("initial_vector_size", i64),
        ("tridiag_mode", i32),
        ("orth_mode", i32),
        ("init_vector", INIT_FN),
        ("init_user", vp),
        ("init_vector_dev", vp),
    ]


class CsrOptions(C.Structure):
    _fields_ = [("accuracy", i32), ("kernel", i32), ("arrays_on_device", i32), ("reserved", i32 * 5)]


class ExpoParams(C.Structure):
    _fields_ = [
        ("matrix_size", i64),
        ("max_iteration", i64),
        ("eps", f64),
        ("full_orthogonalize", i32),
        ("orth_mode", i32),
        ("initial_vector_size", i64),
    ]


class StencilDesc(C.Structure):
    """ll_stencil_desc"""

    _fields_ = [
        ("ndim", C.c_int32),
        ("periodic", C.c_int32 * 3),
        ("dims", C.c_int64 * 3),
        ("diag", C.c_double),
        ("hop_re", C.c_double * 3),
        ("hop_im", C.c_double * 3),
        ("phase_grad", (C.c_double * 3) * 3),
    ]


class RunStats(C.Structure):
    _fields_ = [
        ("n_passes", i64),
        ("total_iterations", i64),
        ("seconds_total", f64),
        ("seconds_host_tridiag", f64),
        ("seconds_spmv", f64),
        ("seconds_orth", f64),
        ("last_alpha_len", i64),
        ("seconds_host_enqueue", f64),
        ("seconds_host_wait", f64),
        ("seconds_setup", f64),
        ("seconds_finish", f64),
        ("second_passes", i64),
        ("seconds_comm_gather", f64),
        ("seconds_comm_allreduce", f64),
        ("lagged_iterations", i64),
        ("pair_iterations", i64),
        ("pair_gate_trips", i64),
        ("reserved", i64 * 6),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if k != "reserved"}


P = C.POINTER
# name -> (restype, argtypes); every symbol of include/lanczos_hip.h
PROTOTYPES = {
    "ll_last_error": (C.c_char_p, []),
    "ll_version": (C.c_int, []),
    "ll_abi_check": (C.c_int, [C.c_int, C.c_int, C.c_size_t, C.c_size_t]),
    "ll_ctx_create": (C.c_int, [C.c_int, P(vp)]),
    "ll_ctx_create_on_stream": (C.c_int, [C.c_int, vp, P(vp)]),
    "ll_ctx_destroy": (C.c_int, [vp]),
    "ll_ctx_stream": (C.c_int, [vp, P(vp)]),
    "ll_ctx_synchronize": (C.c_int, [vp]),
    "ll_ctx_release_cache": (C.c_int, [vp]),
    "ll_ctx_reload_env": (C.c_int, [vp]),
    "ll_ctx_set_tuning": (C.c_int, [vp, C.c_char_p, C.c_char_p]),
    "ll_ctx_set_profiling": (C.c_int, [vp, C.c_int]),
    "ll_timer_start": (C.c_int, [vp]),
    "ll_timer_stop": (C.c_int, [vp, P(f64)]),
    "ll_bandwidth_probe": (C.c_int, [vp, C.c_size_t, P(f64), P(f64)]),
    "ll_comm_unique_id": (C.c_int, [vp]),
    "ll_comm_init": (C.c_int, [vp, vp, C.c_int, C.c_int]),
    "ll_comm_rank": (C.c_int, [vp, P(C.c_int), P(C.c_int)]),
    "ll_comm_ranks_seen": (C.c_int, [vp, P(C.c_int)]),
    "ll_comm_transport": (C.c_int, [vp, C.c_char_p, C.c_size_t]),
    "ll_comm_attach": (C.c_int, [vp, vp, C.c_int, C.c_int]),
    "ll_partition": (C.c_int, [i64, C.c_int, C.c_int, P(i64), P(i64)]),
    "ll_malloc": (C.c_int, [vp, C.c_size_t, P(vp)]),
    "ll_free": (C.c_int, [vp, vp]),
    "ll_memcpy_h2d": (C.c_int, [vp, vp, vp, C.c_size_t]),
    "ll_memcpy_d2h": (C.c_int, [vp, vp, vp, C.c_size_t]),
    "ll_memset": (C.c_int, [vp, vp, C.c_int, C.c_size_t]),
    "ll_op_create_csr_d": (C.c_int, [vp, i64, i64, i64, vp, vp, vp, P(vp)]),
    "ll_op_create_csr_z": (C.c_int, [vp, i64, i64, i64, vp, vp, vp, P(vp)]),
    "ll_op_create_csr_dev_d": (C.c_int, [vp, i64, i64, i64, vp, vp, vp, P(vp)]),
    "ll_op_create_csr_dev_z": (C.c_int, [vp, i64, i64, i64, vp, vp, vp, P(vp)]),
    "ll_csr_options_default": (C.c_int, [P(CsrOptions)]),
    "ll_op_create_csr_opt_d": (C.c_int, [vp, i64, i64, i64, vp, vp, vp, P(CsrOptions), P(vp)]),
    "ll_op_create_csr_opt_z": (C.c_int, [vp, i64, i64, i64, vp, vp, vp, P(CsrOptions), P(vp)]),
    "ll_op_set_accuracy": (C.c_int, [vp, C.c_int]),
    "ll_op_accuracy": (C.c_int, [vp, P(C.c_int)]),
    "ll_op_create_coo_d": (C.c_int, [vp, i64, i64, vp, vp, vp, P(vp)]),
    "ll_op_create_coo_z": (C.c_int, [vp, i64, i64, vp, vp, vp, P(vp)]),
    "ll_op_inf_norm": (C.c_int, [vp, P(f64)]),
    "ll_op_create_dense_d": (C.c_int, [vp, i64, i64, i64, vp, P(vp)]),
    "ll_op_create_dense_z": (C.c_int, [vp, i64, i64, i64, vp, P(vp)]),
    "ll_op_create_stencil_d": (C.c_int, [vp, P(StencilDesc), i64, i64, vp, P(vp)]),
    "ll_op_create_stencil_z": (C.c_int, [vp, P(StencilDesc), i64, i64, vp, P(vp)]),
    "ll_op_create_host_d": (C.c_int, [vp, i64, HOST_MV_FN, vp, P(vp)]),
    "ll_op_create_host_z": (C.c_int, [vp, i64, HOST_MV_FN, vp, P(vp)]),
    "ll_op_create_device_d": (C.c_int, [vp, i64, DEV_MV_FN, vp, P(vp)]),
    "ll_op_create_device_z": (C.c_int, [vp, i64, DEV_MV_FN, vp, P(vp)]),
    "ll_op_select_spmv": (C.c_int, [vp, C.c_int]),
    "ll_op_selected_spmv": (C.c_int, [vp, P(C.c_int)]),
    "ll_op_autotune_ms": (C.c_int, [vp, P(f64), P(f64)]),
    "ll_op_autotune_ms_of": (C.c_int, [vp, C.c_int, P(f64)]),
    "ll_op_tiled_layout": (C.c_int, [vp, P(C.c_int), P(C.c_int)]),
    "ll_op_destroy": (C.c_int, [vp]),
    "ll_op_info": (C.c_int, [vp, P(i64), P(i64), P(i64)]),
    "ll_spmv_d": (C.c_int, [vp, vp, vp, vp, f64, P(f64)]),
    "ll_spmv_z": (C.c_int, [vp, vp, vp, vp, f64, P(f64)]),
    "ll_dot_d": (C.c_int, [vp, i64, vp, vp, P(f64)]),
    "ll_dot_z": (C.c_int, [vp, i64, vp, vp, P(f64)]),
    "ll_nrm2_d": (C.c_int, [vp, i64, vp, P(f64)]),
    "ll_nrm2_z": (C.c_int, [vp, i64, vp, P(f64)]),
    "ll_scal_d": (C.c_int, [vp, i64, f64, vp]),
    "ll_scal_z": (C.c_int, [vp, i64, f64, vp]),
    "ll_normalize_d": (C.c_int, [vp, i64, vp, P(f64)]),
    "ll_normalize_z": (C.c_int, [vp, i64, vp, P(f64)]),
    "ll_three_term_d": (C.c_int, [vp, i64, vp, vp, vp, f64, f64]),
    "ll_three_term_z": (C.c_int, [vp, i64, vp, vp, vp, f64, f64]),
    "ll_orth_block_d": (C.c_int, [vp, i64, i64, vp, i64, vp, C.c_int, P(f64), vp]),
    "ll_orth_block_z": (C.c_int, [vp, i64, i64, vp, i64, vp, C.c_int, P(f64), vp]),
    "ll_gemv_basis_d": (C.c_int, [vp, i64, i64, vp, i64, i64, vp, vp, i64]),
    "ll_gemv_basis_z": (C.c_int, [vp, i64, i64, vp, i64, i64, vp, vp, i64]),
    "ll_tridiag_eig": (C.c_int, [i64, vp, vp, vp, vp, P(i64)]),
    "ll_tridiag_bisect": (C.c_int, [i64, vp, vp, i64, P(f64)]),
    "ll_tridiag_bisect_multi": (C.c_int, [i64, vp, vp, i64, vp, vp]),
    "ll_tridiag_eigvecs": (C.c_int, [i64, vp, vp, i64, vp, vp]),
    "ll_lanczos_params_default": (C.c_int, [P(LanczosParams), i64, C.c_int, i64]),
    "ll_expo_params_default": (C.c_int, [P(ExpoParams), i64]),
    "ll_lanczos_run_d": (C.c_int, [vp, vp, P(LanczosParams), vp, vp, P(i64), vp, i64, vp, vp, P(RunStats)]),
    "ll_lanczos_run_z": (C.c_int, [vp, vp, P(LanczosParams), vp, vp, P(i64), vp, i64, vp, vp, P(RunStats)]),
    "ll_lanczos_run_iteration_d": (C.c_int, [vp, vp, P(LanczosParams), i64, i64, vp, vp, vp, P(i64), P(i64), vp, vp,
                                             P(RunStats)]),
    "ll_lanczos_run_iteration_z": (C.c_int, [vp, vp, P(LanczosParams), i64, i64, vp, vp, vp, P(i64), P(i64), vp, vp,
                                             P(RunStats)]),
    "ll_expo_run_d": (C.c_int, [vp, vp, P(ExpoParams), f64, vp, vp, P(i64), P(RunStats)]),
    "ll_expo_run_z": (C.c_int, [vp, vp, P(ExpoParams), f64, f64, vp, vp, P(i64), P(RunStats)]),
    "ll_expo_taylor_run_d": (C.c_int, [vp, vp, P(ExpoParams), f64, vp, vp, P(i64)]),
    "ll_expo_taylor_run_z": (C.c_int, [vp, vp, P(ExpoParams), f64, f64, vp, vp, P(i64)]),
}

# float storage types: _s mirrors _d, _c mirrors _z (data pointers are void* here, scalars stay double)
for _name in list(PROTOTYPES):
    if _name.endswith("_d") and _name != "ll_memcpy_h2d":
        PROTOTYPES[_name[:-2] + "_s"] = PROTOTYPES[_name]
    elif _name.endswith("_z"):
        PROTOTYPES[_name[:-2] + "_c"] = PROTOTYPES[_name]

_lib = None


def lib():
    """The loaded liblanczos_hip.so (loads on first use; raises if it has not been built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise LanczosHipError(
                LL_ERR_HIP,
                "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU fallback)" % LIB_PATH,
            )
        handle = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        # ABI handshake (lanczos_hip.h, ll_abi_check): the ctypes mirrors below must lay the structs out like the library
        if handle.ll_abi_check(ABI_VERSION[0], ABI_VERSION[1], C.sizeof(RunStats), C.sizeof(LanczosParams)) != LL_OK:
            raise LanczosHipError(LL_ERR_INVALID, handle.ll_last_error().decode("utf-8", "replace"))
        _lib = handle
    return _lib


def check(status):
    if status != LL_OK:
        raise LanczosHipError(status, lib().ll_last_error().decode("utf-8", "replace"))


def ptr(a):
    """void* of a numpy array (or None)."""
    return None if a is None else a.ctypes.data_as(vp)
