import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


# The suite runs the PRODUCTION defaults (the SpMV image that loses the creation-time timing is released, ...); a test
# that needs another setting asks for it through the `llenv` fixture below.
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


class _LLEnv:
    """Set / delete LL_* environment switches for one test.  The library reads them once per context (ll_ctx_create),
    so every change is followed by ll_ctx_reload_env on all live contexts; the old values come back (and are reloaded)
    when the test ends."""

    def __init__(self):
        self._saved = {}

    @staticmethod
    def _reload():
        import lambda_lanczos_amd as L

        for c in L.live_contexts():
            c.reload_env()

    def setenv(self, name, value):
        self._saved.setdefault(name, os.environ.get(name))
        os.environ[name] = str(value)
        self._reload()

    def delenv(self, name, raising=False):
        if name not in os.environ:
            if raising:
                raise KeyError(name)
            return
        self._saved.setdefault(name, os.environ.get(name))
        del os.environ[name]
        self._reload()

    def undo(self):
        for name, old in self._saved.items():
            if old is None:
                os.environ.pop(name, None)
            else:
                os.environ[name] = old
        self._saved.clear()
        self._reload()


@pytest.fixture
def llenv():
    e = _LLEnv()
    yield e
    e.undo()


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
