import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


# Tests switch SpMV kernels on live operators (ll_op_select_spmv), so both matrix images are kept; the default
# (release the image that lost the creation-time timing) has its own test.
os.environ.setdefault("LL_SPMV_KEEP_BOTH", "1")
# The host-staged multi-rank TEST transport (tests/transport/, built by __graft_entry__.build()).
SHM_TRANSPORT = os.path.join(ROOT, "tests", "transport", "_build", "libll_shm_transport.so")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    # Counting devices does not initialise the GPU on this image (see the environment notes).
    try:
        import torch

        return torch.cuda.device_count() > 0
    except Exception:  # noqa: BLE001
        return False


@pytest.fixture(scope="session")
def ctx():
    """One library context for the whole GPU session (fails loudly if the HIP library or the device is missing)."""
    import lambda_lanczos_amd as L

    c = L.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib

    return oracle_lib.oracle()


@pytest.fixture(scope="session")
def reference():
    import oracle_lib

    if not oracle_lib.have_reference():
        pytest.skip("oracle/_ref/libref.so not present")
    return oracle_lib.reference()
