"""AddressSanitizer + UndefinedBehaviorSanitizer leg for the host-side code of the hot path (the
reference builds its tests with -fsanitize=address, /root/reference/test/CMakeLists.txt:3): the product's tridiagonal
solver (csrc/tridiag_host.cpp), the synthetic generators (csrc/generators.cpp) and the CPU oracle, compiled with
-fsanitize=address,undefined -fno-sanitize-recover=all and driven by tests/sanitize/sanitize_main.cpp.
CPU container only: GPU sanitizers are not available on the pool, and the run is skipped where a GPU is present."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _have_gpu():
    try:
        import torch

        return torch.cuda.device_count() > 0
    except Exception:  # noqa: BLE001
        return False


@pytest.mark.skipif(_have_gpu(), reason="sanitizer leg runs in the CPU container only")
def test_host_code_is_clean_under_asan_and_ubsan():
    d = os.path.join(ROOT, "tests", "sanitize")
    b = subprocess.run(["make", "-s", "-C", d, "all"], capture_output=True, text=True, timeout=600)
    assert b.returncode == 0, b.stdout[-2000:] + b.stderr[-4000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1",
               OMP_NUM_THREADS="2")
    r = subprocess.run([os.path.join(d, "_build", "sanitize_host")], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-6000:]
    assert "sanitize ok" in r.stdout
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr


@pytest.mark.skipif(_have_gpu(), reason="sanitizer leg runs in the CPU container only")
@pytest.mark.parametrize("binary,marker", [("worker_asan", "ERROR: AddressSanitizer"),
                                           ("worker_tsan", "WARNING: ThreadSanitizer")])
def test_helper_thread_host_step_is_clean_under_asan_and_tsan(binary, marker):
    """csrc/ritz_tracker.hpp (RitzTracker, ExpoTracker and the helper thread that runs them) driven the way
    lanczos_run / expo_run drive it — opportunistic and fixed-lag verdict consumption, early destruction — under
    ASan+UBSan and under ThreadSanitizer; the threaded verdicts must equal the inline ones bit for bit."""
    d = os.path.join(ROOT, "tests", "sanitize")
    b = subprocess.run(["make", "-s", "-C", d, "all"], capture_output=True, text=True, timeout=600)
    assert b.returncode == 0, b.stdout[-2000:] + b.stderr[-4000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1",
               TSAN_OPTIONS="halt_on_error=1", OMP_NUM_THREADS="1")
    r = subprocess.run([os.path.join(d, "_build", binary)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-6000:]
    assert "worker sanitize ok" in r.stdout
    assert marker not in r.stderr and "runtime error" not in r.stderr
