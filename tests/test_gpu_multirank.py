"""N > 1 ranks with the REAL kernels: 2 and 3 processes share the test box's single GPU and exchange through a
host-staged TEST transport (tests/transport/shm_transport.cpp, attached through LL_COMM_PLUGIN; the product library
holds no such transport) — RCCL refuses several ranks on one device, and the pool has 1-GPU boxes only.  Everything
except the transport is the production sharded path, including the overlapped exchange (all-gather in chunks on the
communication stream, own-column SpMV work under it):
ll_partition row shards, global column indices, the padded all-gather before each SpMV (CSR-stream and propagation
blocking), all-reduced alpha / Gram-Schmidt coefficients / norms, replicated host decisions, sharded locked vectors in
restart passes, sharded Exponentiator input/output."""
import json
import os
import subprocess
import sys
import uuid

import numpy as np
import pytest

import lambda_lanczos_amd as L
from lambda_lanczos_amd import generators as G
from conftest import SHM_TRANSPORT
from util import list2c, overlap

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_ranks(tmp_path, world, **env_extra):
    """Start `world` rank processes on the box's single GPU (test transport) and return their result records."""
    name = "/ll_shm_test_" + uuid.uuid4().hex[:12]
    env = dict(os.environ, LL_COMM_PLUGIN=SHM_TRANSPORT, OMP_NUM_THREADS="2", **env_extra)
    os.makedirs(tmp_path, exist_ok=True)
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "shm_rank_worker.py"), str(r), str(world), name,
                               str(tmp_path)], env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(world)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-4000:]
    return [json.load(open(os.path.join(tmp_path, "rank%d.json" % r))) for r in range(world)]


def test_overlapped_exchange_is_bit_identical_to_the_serial_issue_order(tmp_path):
    """The same 2-rank job with the all-gather on the communication stream (chunked, own-column SpMV under it) and
    with everything on one stream (LL_COMM_OVERLAP=0): every replicated scalar, every Ritz vector shard and the plain
    SpMV results must agree bit for bit — the overlap changes WHEN things run, never what is summed in which order."""
    # LL_SPMV_KERNEL=pb: no timing-based kernel choice (it may differ from run to run); both images are kept (conftest)
    a = run_ranks(os.path.join(tmp_path, "overlap"), 2, LL_COMM_OVERLAP="1", LL_GATHER_CHUNKS="3", LL_SPMV_KERNEL="pb")
    b = run_ranks(os.path.join(tmp_path, "serial"), 2, LL_COMM_OVERLAP="0", LL_GATHER_CHUNKS="3", LL_SPMV_KERNEL="pb")
    for ra, rb in zip(a, b):
        for key in ("randsym_csr", "randsym_pb"):
            assert ra[key]["alpha"] == rb[key]["alpha"] and ra[key]["vals"] == rb[key]["vals"], key
            assert ra[key]["vecs"] == rb[key]["vecs"], key
        assert ra["spmv_pb"] == rb["spmv_pb"] and ra["spmv_csr"] == rb["spmv_csr"]
        assert ra["laplace"] == rb["laplace"]
        assert ra["tiled"] == rb["tiled"]     # the tiled kernel's two launches: under the gather / behind it on one stream


# The CSR-stream and dense operators of a sharded context multiply the rank's OWN columns under the all-gather and add the
# other ranks' columns when the vector has arrived (column-split image, capi.cpp build_csr_split; every case below runs it);
# "csr-gather-then-multiply" (LL_CSR_SPLIT=0) is the round-3 form: gather first, then one kernel over whole rows.
# forced second pass: every iteration takes the host-decided second Gram-Schmidt pass (drain, repeat, re-enqueue): the
# replicated decision must keep the ranks' collective sequences aligned, results unchanged up to rounding.
# measured norm: the post-pass norm from maxpy's partial sums + an all-reduce instead of ||w||^2 - sum |h_j|^2.
# verdict jitter: the helper thread of every rank delivers its stop verdicts up to 3 ms late, at times that differ from
# rank to rank (3 ms >> one iteration here): the ranks must still enqueue the same iterations, i.e. consume verdicts at
# a fixed lag (StepWorker::consume) — with opportunistic consumption a rank that sees the stop earlier leaves its peers
# alone in a collective and the job hangs.  lag 0: every verdict awaited before the next iteration is enqueued.
# one-off stress runs: LL_MULTIRANK_WORLDS=5,6,8 adds plain cases with those rank counts
_EXTRA_WORLDS = [int(w) for w in os.environ.get("LL_MULTIRANK_WORLDS", "").split(",") if w.strip()]


@pytest.mark.parametrize("world,extra", [(w, {}) for w in _EXTRA_WORLDS] + [(2, {}), (3, {}), (4, {}), (2, {"LL_DGKS_THRESHOLD": "2.0"}),
                                         (2, {"LL_SHARDED_NORM": "measured"}), (3, {"LL_GATHER_CHUNKS": "1"}),
                                         (3, {"LL_TRIDIAG_TEST_JITTER_US": "3000"}),
                                         (2, {"LL_TRIDIAG_TEST_JITTER_US": "1500", "LL_TRIDIAG_LAG": "0"}),
                                         (2, {"LL_BLAS_SMALL_BYTES": "0"}), (3, {"LL_BLAS_SMALL_BYTES": "0"}),
                                         (2, {"LL_BLAS_SMALL_BYTES": "0", "LL_DGKS_THRESHOLD": "2.0"}),
                                         (3, {"LL_BLAS_SMALL_BYTES": "0", "LL_TRIDIAG_TEST_JITTER_US": "3000"}),
                                         (2, {"LL_TEST_LAGGED_MIN_BYTES": "0"}), (3, {"LL_TEST_LAGGED_MIN_BYTES": "0"}),
                                         (3, {"LL_CSR_SPLIT": "0"})],
                         ids=["stress-%d" % w for w in _EXTRA_WORLDS] + ["2", "3", "4", "2-forced-second-pass", "2-measured-norm", "3-one-chunk",
                              "3-verdict-jitter", "2-verdict-jitter-lag0", "2-one-sweep", "3-one-sweep",
                              "2-one-sweep-forced-second-pass", "3-one-sweep-verdict-jitter", "2-one-sweep-small-geometry",
                              "3-one-sweep-small-geometry", "3-csr-gather-then-multiply"])
def test_sharded_engine_with_several_ranks_on_one_gpu(tmp_path, oracle, ctx, world, extra, llenv):
    ranks = run_ranks(tmp_path, world, **extra)

    def stitch(key, field, idx=None):
        parts = []
        for r in ranks:
            v = r[key][field]
            parts.append(np.asarray(v if idx is None else v[idx]))
        return np.concatenate(parts)

    # ---- random symmetric, two roots
    n = 9001
    csr = G.randsym_np(n)
    init = G.start_vector(n, 1)
    ora = oracle.lanczos(csr, init, True, num_eigs=2, max_iteration=120)
    y_ref = oracle.spmv(csr, init) + 0.5 * init
    for label in ("csr", "pb"):
        key = "randsym_" + label
        for r in ranks[1:]:   # replicated scalars and decisions
            assert r[key]["vals"] == ranks[0][key]["vals"] and r[key]["iters"] == ranks[0][key]["iters"]
            assert r[key]["alpha"] == ranks[0][key]["alpha"]
        assert sum(r[key]["n_local"] for r in ranks) == n
        # the one-sweep Gram-Schmidt form runs on sharded contexts too (streaming geometry: forced by the *-one-sweep cases)
        one_sweep = extra.get("LL_BLAS_SMALL_BYTES") == "0" or extra.get("LL_TEST_LAGGED_MIN_BYTES") == "0"
        assert all((r[key]["lagged"] > 0) == one_sweep for r in ranks)
        # ... and takes TWO iterations per sweep — in the streaming geometry and, since round 6, in the small-vector geometry
        # (pair_small_kernel) as well: one all-reduce carries both iterations' columns, every rank folds the same numbers
        assert all((r[key]["pair"] > 0) == one_sweep for r in ranks)
        vals = np.array(ranks[0][key]["vals"])
        assert np.max(np.abs(vals - ora["eigenvalues"])) <= 1e-10 * np.max(np.abs(vals))
        assert ranks[0][key]["iters"] == ora["iter_counts"]          # fixed windows: identical pass structure
        m = len(ora["alpha"])
        assert np.max(np.abs(np.array(ranks[0][key]["alpha"])[:m] - ora["alpha"])) <= 1e-10 * 30
        for i in range(2):
            assert 1 - overlap(stitch(key, "vecs", i), ora["eigenvectors"][i]) <= 1e-8
        y = stitch("spmv_" + label, "y")
        assert np.max(np.abs(y - y_ref)) <= 1e-12 * 40
        assert abs(ranks[0]["spmv_" + label]["dot"] - float(init @ y_ref)) <= 1e-9 * n
        if label == "pb" and not os.environ.get("LL_PB_PHASE2"):
            # the fixed-point sums of the PB kernel's phase 2 do not depend on the order of the adds, nor on the block
            # geometry, nor on the partition: the shards of `world` ranks stitch to the BITS of a single-GPU product
            llenv.setenv("LL_SPMV_KERNEL", "pb")
            op1 = L.CsrOperator(ctx, *csr)
            x1, y1 = ctx.to_device(init), ctx.empty(n)
            L.spmv(op1, x1, y1, offset=0.5)
            assert np.array_equal(y, y1.get())
            op1.close()
    # ---- run_iteration with the first eigenvector as a sharded orthogonalizeTo list
    lock = stitch("randsym_csr", "vecs", 0)
    ori = oracle.run_iteration(csr, init, True, 2, orth=lock[None, :], max_iteration=60)
    assert ranks[0]["run_iteration"]["itern"] == ori["itern"]
    got_vals = np.array(ranks[0]["run_iteration"]["vals"])
    assert np.max(np.abs(got_vals - ori["eigenvalues"])) <= 1e-9 * np.max(np.abs(got_vals))
    for i in range(len(got_vals)):
        v = stitch("run_iteration", "vecs", i)
        assert abs(np.vdot(lock, v)) <= 1e-8 and 1 - overlap(v, ori["eigenvectors"][i]) <= 1e-6
    # ---- Laplacian
    lap = G.laplace2d_np(24)
    ora2 = oracle.lanczos(lap, G.start_vector(576, 1), False, offset=-8.0)
    assert abs(ranks[0]["laplace"]["vals"][0] - ora2["eigenvalues"][0]) <= 1e-10 * 8
    assert abs(ranks[0]["laplace"]["iters"][0] - ora2["iter_counts"][0]) <= 2
    assert 1 - overlap(stitch("laplace", "vecs", 0), ora2["eigenvectors"][0]) <= 1e-8
    # ---- complex Exponentiator
    tcsr = G.torus_np(24)
    inp = G.start_vector(576, 1, np.complex128)
    o_out, o_it, _ = oracle.expo(tcsr, -1j, inp)
    out = np.concatenate([list2c(r["torus_expo"]["out"]) for r in ranks])
    assert abs(ranks[0]["torus_expo"]["itern"] - o_it) <= 1
    assert np.max(np.abs(out - o_out)) <= 1e-10 * np.linalg.norm(inp)
    out_t = np.concatenate([list2c(r["torus_expo_tiled"]["out"]) for r in ranks])
    assert abs(ranks[0]["torus_expo_tiled"]["itern"] - o_it) <= 1
    assert np.max(np.abs(out_t - o_out)) <= 1e-10 * np.linalg.norm(inp)
    assert all(r["torus_expo_tiled"]["layout"][0] > 0 for r in ranks)
    # ---- matrix-free lattice operators (halo exchange) and the dense row block
    key = "stencil_laplace"
    for r in ranks[1:]:
        assert r[key]["vals"] == ranks[0][key]["vals"] and r[key]["alpha"] == ranks[0][key]["alpha"]
    assert abs(ranks[0][key]["vals"][0] - ora2["eigenvalues"][0]) <= 1e-10 * 8
    assert abs(ranks[0][key]["iters"][0] - ora2["iter_counts"][0]) <= 2
    m = min(len(ora2["alpha"]), len(ranks[0][key]["alpha"]))
    assert np.max(np.abs(np.array(ranks[0][key]["alpha"])[:m] - ora2["alpha"][:m])) <= 1e-10 * 8
    assert 1 - overlap(stitch(key, "vecs", 0), ora2["eigenvectors"][0]) <= 1e-8
    dims = [9, 5, 7]
    n5 = int(np.prod(dims))
    c5 = G.lattice_csr(dims, diag=0.25, hop=[0.5 + 1j, -1.0, 0.75j], periodic=True, onsite=0.3 * np.cos(np.arange(n5)),
                       dtype=np.complex128)
    x5 = G.start_vector(n5, 3, np.complex128)
    y5 = oracle.spmv(c5, x5) - 0.5 * x5
    got = np.concatenate([list2c(r["stencil_3d"]["y"]) for r in ranks])
    assert np.max(np.abs(got - y5)) <= 1e-13 * 10
    assert abs(ranks[0]["stencil_3d"]["dot"] - np.vdot(x5, y5).real) <= 1e-11 * n5
    o5, it5, _ = oracle.expo(c5, -0.7j, x5)
    out5 = np.concatenate([list2c(r["stencil_3d"]["out"]) for r in ranks])
    assert abs(ranks[0]["stencil_3d"]["itern"] - it5) <= 1
    assert np.max(np.abs(out5 - o5)) <= 1e-10 * np.linalg.norm(x5)
    # ---- banded matrix through the 2-D tiled kernel, sharded: two launches per product, the own-column row blocks under the gather
    n10 = 30011
    band = G.randsym_np(n10, band=700)
    x10 = G.start_vector(n10, 7)
    y10 = oracle.spmv(band, x10) - 0.25 * x10
    for label in ("fixed", "ordered"):
        recs = [r["tiled"][label] for r in ranks]
        assert sum(r["n_local"] for r in recs) == n10
        for r in recs:   # row blocks in the shard's middle need no column of another rank; the ones at its ends do (the band wraps around)
            nrb, own = r["layout"]
            # (from eight ranks on a shard of 3 752 columns holds no whole 2 048-column tile: every row block waits for the gather)
            assert (0 < own if world <= 4 else 0 <= own) and own < nrb, r["layout"]
        y = np.concatenate([np.asarray(r["y"]) for r in recs])
        assert np.max(np.abs(y - y10)) <= 1e-12 * 40, label
        assert abs(recs[0]["dot"] - float(x10 @ y10)) <= 1e-9 * n10
    if not os.environ.get("LL_PB_PHASE2"):
        # fixed-point sums: the shards stitch to the BITS of the single-GPU product — of the tiled kernel and of the PB kernel alike
        y = np.concatenate([np.asarray(r["tiled"]["fixed"]["y"]) for r in ranks])
        llenv.setenv("LL_TL_FORCE", "1")
        for kernel in (L.capi.SPMV_TILED, L.capi.SPMV_PB):
            op1 = L.CsrOperator(ctx, *band, kernel=kernel, accuracy=L.capi.ACCURACY_NORMWISE)
            x1, y1 = ctx.to_device(x10), ctx.empty(n10)
            L.spmv(op1, x1, y1, offset=-0.25)
            assert np.array_equal(y, y1.get()), kernel
            op1.close()
        llenv.delenv("LL_TL_FORCE")
    ora10 = oracle.lanczos(band, G.start_vector(n10, 1), True, num_eigs=2, max_iteration=40)
    t0 = ranks[0]["tiled"]["fixed"]
    for r in ranks[1:]:
        assert r["tiled"]["fixed"]["vals"] == t0["vals"] and r["tiled"]["fixed"]["alpha"] == t0["alpha"]
    assert t0["iters"] == ora10["iter_counts"]
    assert np.max(np.abs(np.array(t0["vals"]) - ora10["eigenvalues"])) <= 1e-10 * np.max(np.abs(ora10["eigenvalues"]))
    m10 = len(ora10["alpha"])
    assert np.max(np.abs(np.array(t0["alpha"])[:m10] - ora10["alpha"])) <= 1e-10 * 30
    for i in range(2):   # (fixed windows of 40 iterations: Ritz vectors of the same Krylov spaces, the second pass with the first one locked)
        v10 = np.concatenate([np.asarray(r["tiled"]["fixed"]["vecs"][i]) for r in ranks])
        assert 1 - overlap(v10, ora10["eigenvectors"][i]) <= 1e-8
    # ---- 2 x 2 problem on `world` ranks (empty shards beyond the second rank)
    assert [r["tiny"]["n_local"] for r in ranks] == [1, 1] + [0] * (world - 2)
    lam = (5 + np.sqrt(5)) / 2
    assert abs(ranks[0]["tiny"]["vals"][0] - lam) <= 1e-12
    v_t = stitch("tiny", "vecs", 0)
    assert np.linalg.norm(np.array([[2.0, 1.0], [1.0, 3.0]]) @ v_t - lam * v_t) <= 1e-12
    dims_v = [12, 6, 8]
    n7 = int(np.prod(dims_v))
    c7 = G.lattice_csr(dims_v, diag=0.25, hop=[0.5 + 1j, -1.0, 0.75j], periodic=[True, False, True],
                       onsite=0.3 * np.cos(np.arange(n7)), dtype=np.complex128,
                       phase_grad=[[0.1, 0.2, 0.3], [0.0, 0.4, -0.2], [0.5, 0.0, 0.7]])
    x7 = G.start_vector(n7, 5, np.complex128)
    y7 = oracle.spmv(c7, x7) + 0.25 * x7
    got7 = np.concatenate([list2c(r["stencil_vec"]["y"]) for r in ranks])
    assert np.max(np.abs(got7 - y7)) <= 1e-13 * 10
    assert abs(ranks[0]["stencil_vec"]["dot"] - np.vdot(x7, y7).real) <= 1e-11 * n7
    rng = np.random.default_rng(5)
    a6 = rng.standard_normal((203, 203))
    a6 = a6 + a6.T
    ora6 = oracle.lanczos(G.dense_to_csr(a6), G.start_vector(203, 1), True)
    assert abs(ranks[0]["dense"]["vals"][0] - ora6["eigenvalues"][0]) <= 1e-10 * np.max(np.abs(ora6["eigenvalues"]))
    assert abs(ranks[0]["dense"]["iters"][0] - ora6["iter_counts"][0]) <= 2
    assert 1 - overlap(stitch("dense", "vecs", 0), ora6["eigenvectors"][0]) <= 1e-8
    a9 = rng.standard_normal((208, 208))      # (the worker draws it from the same generator state)
    a9 = a9 + a9.T
    ora9 = oracle.lanczos(G.dense_to_csr(a9), G.start_vector(208, 1), True)
    assert abs(ranks[0]["dense_aligned"]["vals"][0] - ora9["eigenvalues"][0]) <= 1e-10 * np.max(np.abs(ora9["eigenvalues"]))
    assert abs(ranks[0]["dense_aligned"]["iters"][0] - ora9["iter_counts"][0]) <= 2
    assert 1 - overlap(stitch("dense_aligned", "vecs", 0), ora9["eigenvectors"][0]) <= 1e-8


@pytest.mark.parametrize("extra,kernel", [([], None),
                                          (["--workload", "c3band", "--tuning", "tl_force=1", "--tuning", "spmv_kernel=tiled"], "tl_")],
                         ids=["config4-shape", "banded-through-the-tiled-kernel"])
def test_bench_two_ranks_on_one_gpu(tmp_path, extra, kernel):
    """bench.py exactly as the driver launches it for N = 2 (torch.distributed.run, gloo control plane, id broadcast,
    row shards, sharded SpMV timing, timed Lanczos windows, reductions over ranks, one JSON line from rank 0) — with
    both ranks on the test box's single GPU through the host-staged test transport."""
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, LL_COMM_PLUGIN=SHM_TRANSPORT, LL_BENCH_DEVICE="0", OMP_NUM_THREADS="4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--size", "300000", "--window", "30", "--spmv-reps", "3"] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    d = json.loads(lines[-1])                       # the JSON line is the LAST line of stdout
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "strong"
    assert d["config"]["n"] == 300000 and d["config"]["nnz"] == 15 * 300000
    assert d["value"] > 0 and abs(d["config"]["iterations_per_step"] - 30) < 1e-9
    assert d["cpu_baseline"] is None                # rank 0 at N = 1 only
    if kernel:   # the sharded form of the 2-D tiled kernel ran (tl_force: a band of +-65536 columns on 150 000 rows per rank is not eligible)
        assert kernel in d["spmv"]["kernel"], d["spmv"]["kernel"]
