"""CPU-side checks of the product library: it loads, exports every symbol include/lanczos_hip.h declares, its
host-only pieces (tridiagonal solver, partitioning, parameter defaults) are right, and anything that needs the
device FAILS LOUDLY instead of falling back to a CPU path.  No device compute here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import lambda_lanczos_amd as L
from lambda_lanczos_amd import _capi as capi
from util import load_golden

EPS = np.finfo(np.float64).eps


def _no_gpu():
    import torch

    return torch.cuda.device_count() == 0


def test_library_exports_every_declared_symbol():
    import glob

    text = ""
    for hdr in sorted(glob.glob(os.path.join(os.path.dirname(capi.HEADER_PATH), "*.h"))):  # every include/*.h
        with open(hdr) as f:
            text += f.read()
    declared = set(re.findall(r"^\s*(?:int|const char\*)\s+(ll_[a-z0-9_]+)\s*\(", text, flags=re.M))
    declared -= {"ll_transport_unique_id", "ll_transport_open"}  # exported BY a transport plug-in, not by the library
    assert len(declared) >= 50
    lib = capi.lib()
    for name in sorted(declared):
        assert hasattr(lib, name), name
    # and the ctypes table binds exactly that set
    assert declared == set(capi.PROTOTYPES), declared ^ set(capi.PROTOTYPES)
    assert lib.ll_version() == capi.ABI_VERSION[0] * 1000 + capi.ABI_VERSION[1] == 5


def test_abi_handshake_refuses_a_caller_built_against_another_header():
    """ll_abi_check (lanczos_hip.h): a binary compiled against an older header sees smaller structs than the library fills —
    it must be told so instead of being overrun (the C++ facade and this binding call it once per process)."""
    import ctypes as C

    lib = capi.lib()
    # minors 4 and 5 only added entry points: callers built against minor 3 (same structs), 4 and 5 are all accepted,
    # a caller built against a NEWER header than the loaded library is not
    assert lib.ll_abi_check(0, 3, C.sizeof(capi.RunStats), C.sizeof(capi.LanczosParams)) == capi.LL_OK
    assert lib.ll_abi_check(0, 4, C.sizeof(capi.RunStats), C.sizeof(capi.LanczosParams)) == capi.LL_OK
    assert lib.ll_abi_check(0, 5, C.sizeof(capi.RunStats), C.sizeof(capi.LanczosParams)) == capi.LL_OK
    assert lib.ll_abi_check(0, 6, C.sizeof(capi.RunStats), C.sizeof(capi.LanczosParams)) == capi.LL_ERR_INVALID
    assert lib.ll_abi_check(0, 2, C.sizeof(capi.RunStats) - 72, C.sizeof(capi.LanczosParams)) == capi.LL_ERR_INVALID
    assert b"rebuild the caller" in lib.ll_last_error()


def test_the_library_reads_only_the_documented_environment_switches():
    """A drop-in must not change its numerics path because a stray variable is set: the shipped library names only the
    user-facing switches of INTEGRATION.md section 8 (first table) — the test hooks and geometry overrides are per-context
    settings behind ll_ctx_set_tuning, and no LL_TEST_* name exists in the binary at all."""
    with open(capi.LIB_PATH, "rb") as f:
        blob = f.read()
    assert b"LL_TEST" not in blob
    named = set(m.decode() for m in re.findall(rb"\bLL_[A-Z][A-Z0-9_]{2,}\b", blob))
    root = os.path.dirname(os.path.dirname(capi.HEADER_PATH))
    with open(os.path.join(root, "INTEGRATION.md")) as f:
        text = f.read()
    table = text[text.index("## 8. Environment switches"):]
    table = table[:table.index("Per-context settings")]
    documented = set(re.findall(r"`(LL_[A-Z0-9_]+)", table))
    from util import HOOK_KEYS

    assert not (named & set(HOOK_KEYS)), named & set(HOOK_KEYS)   # none of the harness's hook names is an environment switch
    env_like = {n for n in named if not n.startswith(("LL_ERR", "LL_OK", "LL_SPMV_CSR", "LL_SPMV_PB", "LL_SPMV_TILED", "LL_ORTH",
                                                      "LL_TRIDIAG_QR", "LL_TRIDIAG_BISECT", "LL_TRIDIAG_AUTO", "LL_ACCURACY",
                                                      "LL_VERSION", "LL_ABI", "LL_UNIQUE", "LL_HIP", "LL_REQUIRE", "LL_PB_FIXED",
                                                      "LL_PB_ORDERED", "LL_PB_ATOMIC", "LL_INST"))}
    assert env_like <= documented, env_like - documented


def test_header_cites_the_reference_interface():
    with open(capi.HEADER_PATH) as f:
        text = f.read()
    for cite in ("LL:330-366", "EX:87-173", "LL:120-126", "LA:29-51", "LA:132-144", "TRI:290-361", "LL:133", "EX:175-210"):
        assert cite in text, cite


@pytest.mark.skipif(not _no_gpu(), reason="only meaningful without a device")
def test_no_cpu_fallback_without_device():
    with pytest.raises(L.LanczosHipError) as e:
        L.Context(0)
    assert e.value.code == capi.LL_ERR_HIP and "no CPU fallback" in str(e.value)
    # an engine built on a host callable cannot run either
    with pytest.raises(L.LanczosHipError):
        L.LambdaLanczos(lambda a, b: None, 3, True, 1).run()


def test_product_package_never_touches_the_oracle():
    """The oracle is test infrastructure: no file of the product package may import / dlopen / link it."""
    pkg = os.path.dirname(capi.LIB_PATH.replace(os.sep + "lib" + os.sep, os.sep))
    pkg = os.path.join(os.path.dirname(os.path.dirname(capi.HEADER_PATH)), "lambda-lanczos_amd")
    bad = re.compile(r"oracle_lib|liboracle|libref|oracle/|_ref/")
    for root, _dirs, files in os.walk(pkg):
        if os.sep + "lib" in root:
            continue
        for fn in files:
            if fn.endswith((".py", ".cpp", ".hpp", ".hip", ".h")) or fn == "Makefile":
                with open(os.path.join(root, fn)) as f:
                    for i, line in enumerate(f, 1):
                        assert not bad.search(line), "%s:%d mentions the oracle: %s" % (fn, i, line.strip())
    # the built library has no dynamic dependency on it either
    import subprocess

    out = subprocess.run(["readelf", "-d", capi.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in out and "libref" not in out


def test_params_defaults_are_the_reference_defaults():
    p = capi.LanczosParams()
    capi.check(capi.lib().ll_lanczos_params_default(C.byref(p), 1234, 1, 3))
    assert (p.matrix_size, p.max_iteration, p.find_maximum, p.num_eigs) == (1234, 1234, 1, 3)       # LL:200-208
    assert p.eps == EPS * 1e3 and p.eigenvalue_offset == 0.0                                          # LL:150,165
    assert p.num_eigs_per_iteration == 5 and p.initial_vector_size == 200                             # LL:173,181
    assert p.tridiag_mode == capi.TRIDIAG_AUTO and p.orth_mode == capi.ORTH_CGS_DGKS   # decision-identical to QR
    q = capi.ExpoParams()
    capi.check(capi.lib().ll_expo_params_default(C.byref(q), 77))
    assert (q.matrix_size, q.max_iteration, q.full_orthogonalize, q.initial_vector_size) == (77, 77, 0, 200)
    assert q.eps == EPS * 1e2                                                                         # EX:58
    eng = L.LambdaLanczos.__new__(L.LambdaLanczos)  # python mirror: same defaults without touching the device
    L.LambdaLanczos.__init__(eng, object(), 10, False, 2, dtype=np.float64, context=object())
    assert (eng.max_iteration, eng.eps, eng.num_eigs_per_iteration, eng.initial_vector_size) == (10, EPS * 1e3, 5, 200)


@pytest.mark.parametrize("n,p", [(10, 1), (10, 3), (10_000_000, 8), (7, 8), (1, 2)])
def test_partition_covers_rows_with_equal_strides(n, p):
    shard = -(-n // p)
    seen = 0
    for r in range(p):
        b, c = L.partition(n, p, r)
        assert b == min(n, r * shard) and 0 <= c <= shard
        assert b == seen
        seen += c
    assert seen == n


@pytest.mark.parametrize("name", ["implicit_shift_qr", "null_eigenvalue", "random12", "random40_with_zero_coupling",
                                  "single"])
def test_product_tridiagonal_solver_matches_reference_fixture(name):
    """a11: the product's own flat-array QR reproduces the reference's eigenvalues/eigenvectors (captured fixture)."""
    g = load_golden("tridiagonal.json")[name]
    ev, q, unc = L.tridiag_eig(g["alpha"], g["beta"])
    want_ev, want_q = np.array(g["eigenvalues"]), np.array(g["eigenvectors_rows"])
    scale = max(1.0, np.max(np.abs(want_ev)))
    # same operations in the same order, contraction off on both sides: bit for bit
    assert np.array_equal(ev, want_ev)
    assert np.array_equal(q, want_q)
    assert unc == g["unconverged"]
    for m, want in enumerate(g["bisection"]):
        assert abs(L.tridiag_bisect(g["alpha"], g["beta"], m) - want) <= 4 * EPS * scale


def test_product_tridiagonal_solver_matches_oracle_on_lanczos_matrices(oracle):
    """T_k of real Lanczos runs (clustered Ritz values, tiny couplings): same values as the oracle, every k."""
    from lambda_lanczos_amd import generators as G

    r = oracle.lanczos(G.laplace2d_np(24), G.start_vector(576), False, offset=-8.0)
    al, be = r["alpha"], r["beta"]
    for k in (1, 2, 3, 10, 57, len(al)):
        ev, q, _ = L.tridiag_eig(al[:k], be[: k - 1])
        oev, oq, _ = oracle.tridiag_eig(al[:k], be[:k])
        assert np.max(np.abs(ev - oev)) <= 8 * EPS * 16
        t = np.diag(al[:k]) + np.diag(be[: k - 1], 1) + np.diag(be[: k - 1], -1)
        assert np.max(np.abs(q @ t @ q.T - np.diag(ev))) <= 1e-12 * 16
        lo = L.tridiag_bisect(al[:k], be[: k - 1], 0)
        assert abs(lo - ev[0]) <= 64 * EPS * 16


def test_inverse_iteration_matches_qr_vectors(oracle):
    """The O(m) inverse-iteration eigenvectors (LL_TRIDIAG_AUTO, m > 256) against the QR vectors on the T_m of a
    converged Lanczos run (clustered interior Ritz values, well separated extreme ones) and on a degenerate case."""
    from lambda_lanczos_amd import generators as G

    r = oracle.lanczos(G.laplace2d_np(40), G.start_vector(1600), False, offset=-8.0)
    al, be = r["alpha"], r["beta"]
    m = len(al)
    assert m > 100
    ev, q, _ = L.tridiag_eig(al, be[: m - 1])
    for idx in ([0, 1, 2, 3, 4], [m - 1, m - 2, m - 3, m - 4, m - 5]):
        vecs = L.tridiag_eigvecs(al, be[: m - 1], ev[idx])
        for row, i in zip(vecs, idx):
            assert abs(np.linalg.norm(row) - 1) <= 1e-14
            assert 1 - abs(row @ q[i]) <= 1e-12, (i, 1 - abs(row @ q[i]))
            t_row = al * row
            t_row[:-1] += be[: m - 1] * row[1:]
            t_row[1:] += be[: m - 1] * row[:-1]
            assert np.linalg.norm(t_row - ev[i] * row) <= 1e-12 * 16
    # exactly repeated eigenvalues (two uncoupled copies of the same block): the vectors must still be orthonormal
    a2 = np.array([1.0, 2.0, 3.0, 1.0, 2.0, 3.0])
    b2 = np.array([2.0, 2.0, 0.0, 2.0, 2.0])
    v = L.tridiag_eigvecs(a2, b2, [-1.0, -1.0])
    assert np.max(np.abs(v @ v.T - np.eye(2))) <= 1e-10


import subprocess  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
C_SRC = os.path.join(ROOT, "tests", "c", "cabi_c99.c")
C_EXE = os.path.join(ROOT, "tests", "cpp", "_build", "cabi_c99")


def _build_c99():
    os.makedirs(os.path.dirname(C_EXE), exist_ok=True)
    lib_dir = os.path.join(ROOT, "lambda-lanczos_amd", "lib")
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"), C_SRC,
                    "-o", C_EXE, "-L" + lib_dir, "-llanczos_hip", "-lm", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib"],
                   check=True, capture_output=True, text=True)


def test_header_is_plain_c99_and_links_from_c():
    """include/lanczos_hip.h compiles as strict C99 and a C program links the library: the boundary is a C ABI."""
    _build_c99()
    import torch

    if torch.cuda.device_count() == 0:
        r = subprocess.run([C_EXE], capture_output=True, text=True)
        assert r.returncode == 2 and "no CPU fallback" in r.stdout


@pytest.mark.gpu
def test_c99_program_runs_readme_sample():
    _build_c99()
    r = subprocess.run([C_EXE], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "lambda_max = 4.0000" in r.stdout, r.stdout + r.stderr


def test_multi_root_bisection_is_bit_identical_to_the_single_root_routine():
    """ll_tridiag_bisect_multi interleaves the Sturm recurrences of several roots; every root must follow exactly the
    midpoint sequence of ll_tridiag_bisect (TRI:22-88), including matrices with vanishing couplings."""
    rng = np.random.default_rng(5)
    for m in (1, 2, 3, 9, 64, 65, 257, 1200):
        al = 3.0 + rng.standard_normal(m)
        be = np.abs(1.0 + 0.4 * rng.standard_normal(m))
        if m > 4:
            be[m // 2] = 0.0
        ks = np.unique(np.array([0, 1, 2, 3, 4, m - 1, m - 2, m // 2, m // 3, m // 5, 5 % m, 7 % m]) % m).astype(np.int64)
        out = np.zeros(len(ks))
        capi.check(capi.lib().ll_tridiag_bisect_multi(m, capi.ptr(al), capi.ptr(be), len(ks), capi.ptr(ks), capi.ptr(out)))
        for k, got in zip(ks, out):
            one = C.c_double()
            capi.check(capi.lib().ll_tridiag_bisect(m, capi.ptr(al), capi.ptr(be), int(k), C.byref(one)))
            assert one.value == got, (m, int(k))
        ev = np.linalg.eigvalsh(np.diag(al) + np.diag(be[:-1], 1) + np.diag(be[:-1], -1)) if m > 1 else al
        assert np.max(np.abs(out - ev[ks])) <= 1e-12 * max(1.0, np.max(np.abs(ev)))
