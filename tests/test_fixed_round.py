"""lambda-lanczos_amd/csrc/fixed_round.hpp — the fixed-point SpMV kernels' double -> 64-bit integer rounding in four additions —
against (long long)rint(v) on the host, value by value (every binade, ties, word boundaries, the top of the range).  The header
is the one the kernels include; its additions are the same IEEE operations on the host and on the device."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fixed_round_equals_rint_on_27_million_values(tmp_path):
    exe = str(tmp_path / "fixed_round_test")
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-I", os.path.join(ROOT, "lambda-lanczos_amd", "csrc"),
                    os.path.join(ROOT, "tests", "cpp", "fixed_round_test.cpp"), "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:]
    assert "0 mismatches" in r.stdout
