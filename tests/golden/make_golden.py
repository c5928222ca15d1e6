#!/usr/bin/env python3
"""Generate tests/golden/*.json from the REAL reference (oracle/_ref/libref.so = the reference's own headers compiled
in place by oracle/Makefile).  Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

The fixtures are data — inputs (or the recipe + seed of a generator in lambda-lanczos_amd/generators.py) and the
outputs the reference produced for them.  No reference source is stored.  Cases:
  reference_tests.json   the reference's own test problems (T1/T2) with its own start vector (mt19937(1), T1:25-45)
  tridiagonal.json       T1:757-801 inputs and the reference's tridiagonal_eigenpairs outputs (+ random cases)
  traces.json            alpha/beta traces, iteration counts, Ritz pairs on the SURVEY 8d generators (splitmix start)
  exponentiator.json     T2 problems + torus 32x32 (config 5 in small)
  run_iteration.json     LambdaLanczos::run_iteration called directly (LL:216-322): nroot pairs, caller's orthogonalizeTo
  long_runs.json         runs of several hundred iterations to convergence / exhaustion (the Gram-Schmidt forms of the HIP path
                         against the reference's sequential MGS, LL:260): alpha/beta, counts, values, sampled vector entries

    python tests/golden/make_golden.py run_iteration      # regenerate one file only
    python tests/golden/make_golden.py long_runs          # (about 6 minutes of single-core reference time)
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import cases  # noqa: E402
import oracle_lib  # noqa: E402
from lambda_lanczos_amd import generators as G  # noqa: E402
from util import c2list  # noqa: E402


def dump(name, obj):
    path = os.path.join(HERE, name)
    with open(path, "w") as f:
        json.dump(obj, f, indent=0, separators=(",", ":"))
    print("wrote", path, os.path.getsize(path), "bytes")


def run_iteration_cases():
    """name -> (csr, init, find_maximum, nroot, orth rows or None, offset): shared with the parity tests."""
    import scipy.sparse as sp

    def eigvecs(csr, idx):
        n = csr[0].shape[0] - 1
        w, v = np.linalg.eigh(sp.csr_matrix((csr[2], csr[1], csr[0]), shape=(n, n)).toarray())
        return np.ascontiguousarray(v[:, idx].T)

    c = {}
    m8 = G.dense_to_csr(cases.M8)
    c["m8_three_roots"] = (m8, G.start_vector(8, 1), False, 3, None, 0.0)
    c["m8_lowest_locked"] = (m8, G.start_vector(8, 1), False, 2, eigvecs(m8, [0]), 0.0)
    rs = G.randsym_np(600)
    c["randsym600_top2_locked"] = (rs, G.start_vector(600, 1), True, 3, eigvecs(rs, [-1, -2]), 0.0)
    to = G.torus_np(12)
    c["torus12_lowest_locked"] = (to, G.start_vector(144, 1, np.complex128), False, 2, eigvecs(to, [0]), -10.0)
    return c


def make_run_iteration(ref):
    out = {}
    for name, (csr, init, find_max, nroot, orth, offset) in run_iteration_cases().items():
        r = ref.run_iteration(csr, init, find_max, nroot, orth=orth, offset=offset)
        out[name] = {"nroot": nroot, "find_maximum": find_max, "offset": offset,
                     "orth": None if orth is None else [c2list(v) for v in orth],
                     "eigenvalues": r["eigenvalues"].tolist(), "eigenvectors": [c2list(v) for v in r["eigenvectors"]],
                     "itern": r["itern"]}
    dump("run_iteration.json", out)


def drop_in_ring(n=2000):
    """The matrix of examples/drop_in.cpp: ring with alternating on-site energies, as CSR."""
    rows, cols, vals = [], [], []
    for i in range(n):
        rows += [i, i, i]
        cols += [i, (i + 1) % n, (i + n - 1) % n]
        vals += [0.3 if i % 2 else -0.3, -1.0, -1.0]
    return G.coo_to_csr(n, rows, cols, np.array(vals))


def sample_indices(n, count=512):
    """Fixed, generator-defined positions at which long_runs.json keeps eigenvector entries."""
    return (G.splitmix64(np.arange(count, dtype=np.uint64) + np.uint64(0xABCDEF)) % np.uint64(n)).astype(np.int64)


def long_run_specs():
    """name -> spec; shared with tests/test_gpu_long_runs.py (which rebuilds the matrices from the same generators)."""
    return {
        # Krylov space exhausted at m = 1002 (1002 distinct eigenvalues); the reference stops at 1003 / ~985
        "ring2000_two_lowest_s1": dict(gen="drop_in_ring", args=[2000], find_max=False, offset=-3.0, num_eigs=2, seed=1,
                                       fresh_seed=11),
        "ring2000_two_lowest_s2": dict(gen="drop_in_ring", args=[2000], find_max=False, offset=-3.0, num_eigs=2, seed=2),
        "ring2000_two_lowest_s3": dict(gen="drop_in_ring", args=[2000], find_max=False, offset=-3.0, num_eigs=2, seed=3),
        # Misconvergence of the reference's stopping rule (LL:290-309), found by sweeping 6000 std::mt19937 start vectors
        # (profiles/r04_ring_start_vector_sweep.txt): this vector has an overlap of 2.4e-5 with the ground state, all five
        # tracked Ritz values stand still at m = 1000 before that component has grown, and the reference returns E1 as the
        # "lowest" eigenvalue after 1000 iterations.  A drop-in has to do exactly the same.  (The vector is stored: libstdc++'s
        # uniform_real_distribution over mt19937 is not reproduced by numpy.)
        "ring2000_misconverged_mt1967": dict(gen="drop_in_ring", args=[2000], find_max=False, offset=-3.0, num_eigs=1,
                                             mt19937_seed=1967),
        "randsym1e5_converge": dict(gen="randsym", args=[100000], find_max=True, offset=0.0, num_eigs=1, seed=1),
        "laplace200_converge": dict(gen="laplace2d", args=[200], find_max=False, offset=-8.0, num_eigs=1, seed=1),
        "randsym1e6_fixed120": dict(gen="randsym", args=[1000000], find_max=True, offset=0.0, num_eigs=1, seed=1,
                                    max_iteration=120),
        "randsym1e5_three_roots": dict(gen="randsym", args=[100000], find_max=True, offset=0.0, num_eigs=3, seed=1),
        # Round 5: the STREAMING one-sweep kernel (lagged_kernel, vectors >= 1 MiB) over whole runs.  400 x 400 Laplacian,
        # smallest pair, offset -8: 1.28 MB vectors, ~1400 reference iterations; complex torus 300 x 300 (config 5's matrix in
        # small), smallest pair, offset -10: 1.44 MB vectors, several hundred iterations.
        "laplace400_converge": dict(gen="laplace2d", args=[400], find_max=False, offset=-8.0, num_eigs=1, seed=1),
        "torus300_converge": dict(gen="torus", args=[300], find_max=False, offset=-10.0, num_eigs=1, seed=1, complex=True),
        # Round 5: a run LONGER than one workgroup of the pair sweep holds coefficient columns for (2 497 stored vectors): the sweep
        # splits into two launches by itself.  800 x 800 Laplacian, smallest pair, offset -8: 5.12 MB vectors, 2 557 iterations
        # (the reference takes 85 minutes on one core for this one).
        "laplace800_converge": dict(gen="laplace2d", args=[800], find_max=False, offset=-8.0, num_eigs=1, seed=1),
        # Round 6: BASELINE's configurations AT THEIR FULL SIZES against the real reference (tests/test_gpu_round2.py).  Config 3
        # (random symmetric CSR n = 1e7, nnz = 1.5e8, largest pair) over the 100-iteration window the headline metric is quoted
        # on and with the reference's defaults to convergence (301 iterations, a 24 GB basis on the host); config 2 (1000 x 1000
        # Laplacian, smallest pair, offset -8) over a 200-iteration window.  About 12 minutes of single-core reference time.
        "c3_window100": dict(gen="randsym", args=[10000000], find_max=True, offset=0.0, num_eigs=1, seed=1, max_iteration=100),
        "c3_converge": dict(gen="randsym", args=[10000000], find_max=True, offset=0.0, num_eigs=1, seed=1),
        "c2_window200": dict(gen="laplace2d", args=[1000], find_max=False, offset=-8.0, num_eigs=1, seed=1, max_iteration=200),
    }


def long_run_matrix(spec):
    if spec["gen"] == "drop_in_ring":
        return drop_in_ring(*spec["args"])
    return getattr(G, spec["gen"])(*spec["args"])


def make_long_runs(ref, only=None):
    import time

    out = {}
    path = os.path.join(HERE, "long_runs.json")
    if only and os.path.exists(path):  # regenerate the named entries, keep the others
        with open(path) as f:
            out = json.load(f)
    for name, s in long_run_specs().items():
        if only and name not in only:
            continue
        t0 = time.time()
        csr = long_run_matrix(s)
        n = csr[0].shape[0] - 1
        dtype = np.complex128 if s.get("complex") else np.float64
        init = ref.init_mt19937(s["mt19937_seed"], n) if "mt19937_seed" in s else G.start_vector(n, s["seed"], dtype)
        r = ref.lanczos(csr, init, s["find_max"], num_eigs=s["num_eigs"], offset=s["offset"],
                        max_iteration=s.get("max_iteration"))
        first = r
        if s["num_eigs"] > 1:  # the trace of pass 1 (the shim records the last pass): pass 1 does not depend on num_eigs
            first = ref.lanczos(csr, init, s["find_max"], num_eigs=1, offset=s["offset"], max_iteration=s.get("max_iteration"))
        idx = sample_indices(n)
        out[name] = dict(s, n=n, start="generators.start_vector(n, seed)", iter_counts=r["iter_counts"],
                         eigenvalues=r["eigenvalues"].tolist(), alpha_pass1=first["alpha"].tolist(),
                         beta_pass1=first["beta"][:-1].tolist(),
                         sample="make_golden.sample_indices(n)",
                         eigenvector_samples=[c2list(v[idx]) if s.get("complex") else v[idx].tolist()
                                              for v in r["eigenvectors"]])
        if "mt19937_seed" in s:
            out[name]["start"] = "the reference's own initialiser: std::mt19937(seed) + uniform_real_distribution(-1, 1) (LL:70-104)"
            out[name]["start_vector"] = init.tolist()
        if s.get("fresh_seed"):
            # what run() does by default in a restart pass (LL:334-354 with the std::random_device start of LL:70-104): a
            # FRESH start vector, orthogonalised against the locked pairs.  On this ring every eigenvalue but two is doubly
            # degenerate, so the fresh vector brings the partner of the locked E1 back; the pass runs through the first
            # exhaustion (m = 1001) and on, driven by rounding noise — its iteration COUNT is not reproducible (the reference
            # itself gives 1801..1895 for different vectors), its converged Ritz values are.
            fresh = G.start_vector(n, s["fresh_seed"])
            r2 = ref.run_iteration(csr, fresh, s["find_max"], 5, orth=r["eigenvectors"], offset=s["offset"])
            out[name]["fresh_pass"] = dict(start="generators.start_vector(n, fresh_seed), orthogonalizeTo = the two pairs above",
                                           itern=r2["itern"], eigenvalues=r2["eigenvalues"].tolist())
            print("  fresh pass:", r2["itern"], r2["eigenvalues"], flush=True)
        print(name, "iter_counts", r["iter_counts"], "values", r["eigenvalues"], "%.1f s" % (time.time() - t0), flush=True)
    dump("long_runs.json", out)


def main():
    ref = oracle_lib.reference()
    if len(sys.argv) > 1 and sys.argv[1] == "run_iteration":
        make_run_iteration(ref)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "long_runs":
        make_long_runs(ref, only=sys.argv[2:] or None)
        return
    make_run_iteration(ref)

    # ---- G1/G2/G5/G6: the reference's own eigen tests with its own seeded initializer
    out = {}
    for name, case in cases.eigen_cases().items():
        csr = case["csr"]
        n = csr[0].shape[0] - 1
        dtype = csr[2].dtype
        init = ref.init_mt19937(1, n, dtype)
        r = ref.lanczos(csr, init, case["find_maximum"], num_eigs=case["num_eigs"], eps=case["eps"],
                        offset=case["offset"])
        out[name] = {
            "ref": case["ref"],
            "init_mt19937_seed1": c2list(init),
            "eigenvalues": r["eigenvalues"].tolist(),
            "eigenvectors": [c2list(v) for v in r["eigenvectors"]],
            "iter_counts": r["iter_counts"],
        }
    dump("reference_tests.json", out)

    # ---- G3: tridiagonal
    tri = {}
    rng = np.random.default_rng(2026)
    tcases = {
        "implicit_shift_qr": ([1.0, 2.0, 3.0], [2.0, 2.0]),                                            # T1:758-759
        "null_eigenvalue": ([6.82333617e-03, 3.09398208e00, 1.89919458e00, 1.28531906e-16],
                            [1.19582528e-01, -1.37689656e00, 6.16147405e-15]),                          # T1:787-788
        "random12": (rng.uniform(-2, 2, 12).tolist(), rng.uniform(-1, 1, 11).tolist()),
        "random40_with_zero_coupling": (rng.uniform(-2, 2, 40).tolist(),
                                        [0.0 if i == 17 else float(v) for i, v in enumerate(rng.uniform(-1, 1, 39))]),
        "single": ([2.5], []),
    }
    for name, (al, be) in tcases.items():
        ev, q, unc = ref.tridiag_eig(np.array(al), np.array(be + [0.0]))
        tri[name] = {"alpha": al, "beta": be, "eigenvalues": ev.tolist(), "eigenvectors_rows": q.tolist(),
                     "unconverged": unc,
                     "bisection": [ref.mth_eigenvalue(np.array(al), np.array(be), m) for m in range(len(al))]
                     if len(al) > 1 else []}
    dump("tridiagonal.json", tri)

    # ---- G8: traces on the SURVEY generators, splitmix64 start vector (seed 1)
    tr = {}
    specs = {
        "laplace64_fixed40": dict(gen="laplace2d_np", args=[64], find_max=False, offset=-8.0, max_iteration=40),
        "laplace64_converge": dict(gen="laplace2d_np", args=[64], find_max=False, offset=-8.0, max_iteration=None),
        "randsym4096_converge": dict(gen="randsym_np", args=[4096], find_max=True, offset=0.0, max_iteration=None),
        "torus16_hermitian": dict(gen="torus_np", args=[16], find_max=False, offset=-10.0, max_iteration=None),
    }
    for name, s in specs.items():
        csr = getattr(G, s["gen"])(*s["args"])
        n = csr[0].shape[0] - 1
        init = G.start_vector(n, 1, csr[2].dtype)
        r = ref.lanczos(csr, init, s["find_max"], offset=s["offset"], max_iteration=s["max_iteration"])
        tr[name] = dict(s, n=n, start="generators.start_vector(n, seed=1)",
                        iter_counts=r["iter_counts"], eigenvalues=r["eigenvalues"].tolist(),
                        alpha=r["alpha"].tolist(), beta=r["beta"][:-1].tolist(),
                        note="alpha/beta recovered with the instrumented mv_mul of oracle/ref_shim.cpp",
                        eigenvector=c2list(r["eigenvectors"][0]))
    dump("traces.json", tr)

    # ---- G7: exponentiator
    ex = {}
    for name, case in cases.expo_cases().items():
        o, it, _ = ref.expo(case["csr"], case["a"], case["input"], full_orthogonalize=case["full"])
        ot, terms, _ = ref.expo(case["csr"], case["a"], case["input"], taylor=True)
        ex[name] = {"ref": case["ref"], "a": c2list(np.array([case["a"]], dtype=np.complex128)),
                    "input": c2list(case["input"]), "output": c2list(o), "itern": it, "taylor_output": c2list(ot),
                    "taylor_terms": terms}
    for dt in (0.1, 1.0, 5.0):
        csr = G.torus_np(32)
        inp = G.start_vector(1024, 1, np.complex128)
        o, it, _ = ref.expo(csr, -1j * dt, inp)
        ex["torus32_dt%g" % dt] = {"gen": "torus_np(32)", "start": "generators.start_vector(1024, 1, complex128)",
                                   "a": c2list(np.array([-1j * dt])), "output": c2list(o), "itern": it}
    dump("exponentiator.json", ex)
    make_long_runs(ref)


if __name__ == "__main__":
    main()
