"""Shows why sharded contexts consume the helper thread's verdicts at a fixed lag: 3 ranks on one GPU (test transport),
verdicts delayed by a per-rank pseudo-random time.  LL_TRIDIAG_LAG=-1 (the single-process opportunistic policy) lets the
ranks enqueue different numbers of iterations — the job dies in a collective; the default (lag 3) completes.
  python tools/r02_lockstep_demo.py  > gpurun_out/r02_lockstep_demo.txt"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import test_gpu_multirank as M  # noqa: E402

for lag in ("3", "-1"):
    t0 = time.time()
    try:
        ranks = M.run_ranks(tempfile.mkdtemp(), 3, LL_TRIDIAG_TEST_JITTER_US="3000", LL_TRIDIAG_LAG=lag)
        print("LL_TRIDIAG_LAG=%s: completed in %.1f s, iterations %s" % (lag, time.time() - t0, ranks[0]["randsym_pb"]["iters"]))
    except BaseException as e:  # noqa: BLE001
        print("LL_TRIDIAG_LAG=%s: FAILED after %.1f s: %s" % (lag, time.time() - t0, str(e).strip().splitlines()[-1][:300]))
