"""lambda_lanczos_amd — MI355X-native Lanczos hot path (drop-in for the Krylov loop of mrcdr/lambda-lanczos).

The product is lib/liblanczos_hip.so (hand-written HIP for gfx950 + C++ host drivers behind the C ABI of
include/lanczos_hip.h).  This Python package is the thin host mirror of the reference interface used by the
tests and bench.py; the C++ facade is include/lambda_lanczos_hip/.
"""
from . import _capi as capi  # noqa: F401
from . import generators  # noqa: F401
from ._capi import (  # noqa: F401
    LIB_PATH,
    ORTH_CGS2,
    ORTH_CGS_DGKS,
    ORTH_MGS,
    TRIDIAG_AUTO,
    TRIDIAG_BISECT,
    TRIDIAG_QR,
    LanczosHipError,
)
from .engine import (  # noqa: F401
    Context,
    CsrOperator,
    DenseOperator,
    DeviceArray,
    Exponentiator,
    HostOperator,
    StencilOperator,
    LambdaLanczos,
    default_context,
    live_contexts,
    CONTEXT_CREATED_HOOKS,
    dot,
    gemv_basis,
    normalize,
    nrm2,
    orth_block,
    partition,
    scal,
    spmv,
    three_term,
    tridiag_bisect,
    tridiag_eig,
    tridiag_eigvecs,
)

__version__ = "0.1"
