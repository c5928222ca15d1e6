"""The user-facing examples under examples/ (code written against the reference's API shape, switched over by changing
the include): they must compile with a plain host compiler against the C ABI, and on a GPU they must run and check
themselves (eigenvalues of the host-lambda path = those of the device-resident path; a wave packet evolved in device
memory keeps its norm and moves at the group velocity)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EX = os.path.join(ROOT, "examples")


def build():
    r = subprocess.run(["make", "-s", "-C", EX, "all"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]


def test_examples_compile_with_host_compiler():
    build()
    for name in ("drop_in", "time_evolution"):
        assert os.path.exists(os.path.join(EX, "_build", name))


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["drop_in", "time_evolution"])
def test_examples_run_and_check_themselves(name):
    build()
    # LL_ITER_TRACE: should a run ever fail, the assertion message carries the last iterations of every pass (alpha, beta^2, the
    # two Gram-Schmidt norms, what the user's mv_mul saw and returned) — round 3's one-off failure left no such evidence
    trace = os.path.join(EX, "_build", name + ".trace")
    if os.path.exists(trace):
        os.remove(trace)
    r = subprocess.run([os.path.join(EX, "_build", name)], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, LL_ITER_TRACE=trace))
    ok = r.returncode == 0 and r.stdout.strip().endswith("OK")
    evidence = ""
    if not ok and os.path.exists(trace):
        with open(trace) as f:
            lines = f.read().splitlines()
        stops = [i for i, ln in enumerate(lines) if ln.startswith("stop")]
        evidence = "\n".join("\n".join(lines[max(0, i - 12): i + 1]) for i in stops)[-6000:]
    assert ok, r.stdout[-2000:] + r.stderr[-2000:] + "\n--- iteration trace before every stop ---\n" + evidence
