"""Child process of tests/test_gpu_engines.py::test_context_on_a_torch_stream_with_torch_memory: the library on a stream
owned by torch, working on torch-allocated device memory, ordered with torch kernels on that stream without any
synchronisation in between; then a whole eigen-solve on the same stream."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

import lambda_lanczos_amd as L  # noqa: E402
from util import install_hook_sync  # noqa: E402
import oracle_lib  # noqa: E402
from lambda_lanczos_amd import _capi as capi  # noqa: E402
from lambda_lanczos_amd import generators as G  # noqa: E402


install_hook_sync()   # the harness's hook settings (util.HOOK_KEYS in os.environ) -> every context of this process


def main():
    oracle = oracle_lib.oracle()
    dev = torch.device("cuda:0")
    stream = torch.cuda.Stream(device=dev)
    ctx = L.Context(0, stream=stream.cuda_stream)
    assert ctx.stream() == stream.cuda_stream
    n = 30011
    csr = G.randsym_np(n)
    op = L.CsrOperator(ctx, *csr)
    host_x = G.start_vector(n, 3)
    with torch.cuda.stream(stream):
        x_t = torch.from_numpy(host_x).to(dev) * 2.0                         # device-side work on the shared stream
        y_t = torch.empty(n, dtype=torch.float64, device=dev)
        dot = C.c_double()
        capi.check(capi.lib().ll_spmv_d(ctx.handle, op.handle, C.c_void_p(x_t.data_ptr()), C.c_void_p(y_t.data_ptr()), 0.5,
                                         C.byref(dot)))
        z_t = y_t * 1.0                                                       # consumer on the same stream
    stream.synchronize()
    y_ref = oracle.spmv(csr, 2.0 * host_x) + 0.5 * 2.0 * host_x
    assert np.max(np.abs(z_t.cpu().numpy() - y_ref)) <= 1e-12 * 40
    assert abs(dot.value - float((2.0 * host_x) @ y_ref)) <= 1e-9 * n
    eng = L.LambdaLanczos(op, n, True, 1)
    init = G.start_vector(n, 1)
    eng.init_vector = lambda v, *_: v.__setitem__(slice(None), init)
    vals, vecs = eng.run()
    ora = oracle.lanczos(csr, init, True)
    assert abs(vals[0] - ora["eigenvalues"][0]) <= 1e-10 * abs(vals[0])
    assert eng.getIterationCounts() == ora["iter_counts"]
    op.close()
    ctx.close()
    print("torch stream ok", flush=True)


if __name__ == "__main__":
    main()
