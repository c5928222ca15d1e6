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
    from util import install_hook_sync

    install_hook_sync()   # every Context of this process takes the harness's hook settings (util.HOOK_KEYS)


def _have_gpu():
    # Counting devices does not initialise the GPU on this image (see the environment notes).
    try:
        import torch

        return torch.cuda.device_count() > 0
    except Exception:  # noqa: BLE001
        return False


class _LLEnv:
    """Set / delete LL_* switches for one test.  The library reads the user-facing ones once per context (ll_ctx_create), so
    every change is followed by ll_ctx_reload_env on all live contexts; the test hooks (util.HOOK_KEYS) are not environment
    switches of the library at all — the harness carries them in os.environ (worker processes inherit them) and applies them
    to every context through ll_ctx_set_tuning.  The old values come back when the test ends."""

    def __init__(self):
        self._saved = {}

    @staticmethod
    def _reload():
        import lambda_lanczos_amd as L
        from util import sync_hooks

        for c in L.live_contexts():
            c.reload_env()   # the user-facing switches: the library reads them from the environment
            sync_hooks(c)    # the test hooks and geometry overrides: per-context settings (ll_ctx_set_tuning)

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
