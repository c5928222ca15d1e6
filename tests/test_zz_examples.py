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
    r = subprocess.run([os.path.join(EX, "_build", name)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout[-2000:] + r.stderr[-2000:]
