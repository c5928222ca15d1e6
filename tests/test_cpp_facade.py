"""The C++ drop-in facade (include/lambda_lanczos_hip/*.hpp): user code in the reference's API idiom, checked on the
reference's known-answer problems (tests/cpp/facade_test.cpp).  CPU: the headers compile with a plain host compiler against the C ABI and the program
refuses to run without a device.  GPU: the tests pass."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "facade_test.cpp")
OUT_DIR = os.path.join(ROOT, "tests", "cpp", "_build")
EXE = os.path.join(OUT_DIR, "facade_test")
LIB_DIR = os.path.join(ROOT, "lambda-lanczos_amd", "lib")


def build():
    os.makedirs(OUT_DIR, exist_ok=True)
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"), SRC, "-o", EXE,
           "-L" + LIB_DIR, "-llanczos_hip", "-Wl,-rpath," + LIB_DIR, "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib"]
    subprocess.run(cmd, check=True, capture_output=True, text=True)


def _no_gpu():
    import torch

    return torch.cuda.device_count() == 0


def test_facade_compiles_with_host_compiler():
    build()
    assert os.path.exists(EXE)


@pytest.mark.skipif(not _no_gpu(), reason="only meaningful without a device")
def test_facade_fails_loudly_without_device():
    build()
    r = subprocess.run([EXE], capture_output=True, text=True)
    assert r.returncode == 2 and "no CPU fallback" in r.stdout


@pytest.mark.gpu
def test_reference_tests_through_the_facade():
    build()
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "PASSED" in r.stdout and r.stdout.count("[case]") == 16
