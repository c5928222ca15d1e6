"""Two processes, two DEVICES, the real RCCL communicator: the first multi-device execution of the sharded path is a
checked test, not the bench.  Skipped on boxes with fewer than two GPUs (the pool's boxes have one); the same worker and
the same checks run there over the host-staged test transport with both ranks on device 0, so the harness itself is known
to work."""
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
from util import overlap

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def device_count():
    import torch

    return torch.cuda.device_count()


def run_pair(out_dir, env_extra, real_rccl):
    os.makedirs(out_dir, exist_ok=True)
    if real_rccl:
        uid = L.Context.unique_id().hex()
        env = dict(os.environ, **env_extra)
        env.pop("LL_COMM_PLUGIN", None)
    else:
        uid = ("/ll_shm_rccl2_" + uuid.uuid4().hex[:12]).encode().hex()
        env = dict(os.environ, LL_COMM_PLUGIN=SHM_TRANSPORT, LL_TEST_SAME_DEVICE="1", **env_extra)
    env.setdefault("OMP_NUM_THREADS", "2")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "rccl_rank_worker.py"), str(r), "2", uid, str(out_dir)],
                              env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-4000:]
    return [json.load(open(os.path.join(out_dir, "rank%d.json" % r))) for r in range(2)]


def check_pair(tmp_path, oracle, real_rccl):
    a = run_pair(os.path.join(tmp_path, "overlap"), {"LL_COMM_OVERLAP": "1", "LL_GATHER_CHUNKS": "3"}, real_rccl)
    b = run_pair(os.path.join(tmp_path, "serial"), {"LL_COMM_OVERLAP": "0", "LL_GATHER_CHUNKS": "3"}, real_rccl)
    assert all(r["ranks_seen"] == 2 for r in a + b)
    # 1. overlapped (gather in chunks on the communication stream) == serial issue order, bit for bit
    for ra, rb in zip(a, b):
        for key in ("spmv_pb", "spmv_csr", "lanczos_pb", "lanczos_csr", "torus_expo", "stencil"):
            assert ra[key] == rb[key], key
    # 2. replicated scalars are identical on both ranks
    for key in ("lanczos_pb", "lanczos_csr"):
        assert a[0][key]["vals"] == a[1][key]["vals"] and a[0][key]["alpha"] == a[1][key]["alpha"]
        assert a[0][key]["iters"] == a[1][key]["iters"]
    assert a[0]["stencil"] == a[1]["stencil"] and a[0]["torus_expo"]["itern"] == a[1]["torus_expo"]["itern"]
    # 3. parity with the oracle on the stitched results
    n = 60013
    csr = G.randsym_np(n)
    x = G.start_vector(n, 3)
    y_ref = oracle.spmv(csr, x) + 0.5 * x
    init = G.start_vector(n, 1)
    ora = oracle.lanczos(csr, init, True, num_eigs=2, max_iteration=60)
    for label in ("pb", "csr"):
        y = np.concatenate([np.asarray(r["spmv_" + label]["y"]) for r in a])
        assert np.max(np.abs(y - y_ref)) <= 1e-12 * 40
        assert abs(a[0]["spmv_" + label]["dot"] - float(x @ y_ref)) <= 1e-9 * n
        key = "lanczos_" + label
        vals = np.array(a[0][key]["vals"])
        assert np.max(np.abs(vals - ora["eigenvalues"])) <= 1e-10 * np.max(np.abs(vals))
        assert a[0][key]["iters"] == ora["iter_counts"]
        m = len(ora["alpha"])
        assert np.max(np.abs(np.array(a[0][key]["alpha"])[:m] - ora["alpha"])) <= 1e-10 * 30
        for i in range(2):
            v = np.concatenate([np.asarray(r[key]["vecs"][i]) for r in a])
            assert 1 - overlap(v, ora["eigenvectors"][i]) <= 1e-8
    # the fixed-point PB sums do not depend on the partition: the two shards stitch to the bits of a single-GPU product
    if not os.environ.get("LL_PB_PHASE2"):
        c = L.Context(0)
        os.environ["LL_SPMV_KERNEL"] = "pb"
        try:
            c.reload_env()
            op1 = L.CsrOperator(c, *csr)
            x1, y1 = c.to_device(x), c.empty(n)
            L.spmv(op1, x1, y1, offset=0.5)
            y_pb = np.concatenate([np.asarray(r["spmv_pb"]["y"]) for r in a])
            assert np.array_equal(y_pb, y1.get())
            op1.close()
        finally:
            os.environ.pop("LL_SPMV_KERNEL", None)
            c.close()
    tcsr = G.torus_np(40)
    inp = G.start_vector(1600, 1, np.complex128)
    o_out, o_it, _ = oracle.expo(tcsr, -1j, inp)
    out = np.concatenate([np.asarray(r["torus_expo"]["re"]) + 1j * np.asarray(r["torus_expo"]["im"]) for r in a])
    assert a[0]["torus_expo"]["itern"] == o_it and np.max(np.abs(out - o_out)) <= 1e-10 * np.linalg.norm(inp)
    lap = G.laplace2d_np(48)
    o4 = oracle.lanczos(lap, G.start_vector(48 * 48, 1), False, offset=-8.0, max_iteration=80)
    assert abs(a[0]["stencil"]["vals"][0] - o4["eigenvalues"][0]) <= 1e-10 * 8
    m4 = len(o4["alpha"])
    assert np.max(np.abs(np.array(a[0]["stencil"]["alpha"])[:m4] - o4["alpha"])) <= 1e-10 * 16


@pytest.mark.skipif(device_count() < 2, reason="needs two GPUs (real RCCL refuses two ranks on one device)")
def test_two_devices_real_rccl_overlapped_equals_serial_and_matches_the_oracle(tmp_path, oracle):
    check_pair(tmp_path, oracle, real_rccl=True)


def test_the_same_worker_and_checks_over_the_test_transport_on_one_device(tmp_path, oracle):
    check_pair(tmp_path, oracle, real_rccl=False)
