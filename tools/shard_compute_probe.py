"""Per-rank COMPUTE time of BASELINE config 4 (config 3 row-partitioned over N GPUs), measured on ONE GPU: one process
plays rank 0 of N alone through the measurement stand-in tests/transport/solo_transport.cpp (its "collectives" are
device-local copies / scalings — what they cost is reported separately and is NOT the cost of the real exchange).
Everything else is the production sharded path: the row shard of the matrix with global column indices, the PB image
with own / remote column blocks, the chunked gather buffer, n/N-sized Gram-Schmidt sweeps, all-reduced coefficients.

    python tools/shard_compute_probe.py [N ...]      -> one JSON line per N (default 1 2 4 8)
    SHARD_PROBE_BAND=65536 python tools/shard_compute_probe.py ...   the banded variant of config 3 (columns within +-2^16 of the row):
                                                      eligible for the 2-D tiled kernel, sharded contexts included

Feeds the strong-scaling model of DESIGN.md section 6: T_iter(N) = compute(N) [measured here] + exchange(N) [modelled]."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SOLO = os.path.join(ROOT, "tests", "transport", "_build", "libll_solo_transport.so")

CHILD = r"""
import json, os, sys, time
sys.path.insert(0, %r)
os.environ.setdefault("OMP_WAIT_POLICY", "passive")
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
import numpy as np
import lambda_lanczos_amd as L
from lambda_lanczos_amd import generators as G
N = int(sys.argv[1]); n = 10_000_000; window = 100; band = int(os.environ.get('SHARD_PROBE_BAND', '0'))
ctx = L.Context(0)
if N > 1:
    ctx.init_comm(L.Context.unique_id(), 0, N)
rb, nl = ctx.partition(n)
csr = G.randsym(n, band=band, row_begin=rb, n_local=nl)
nnz_local = int(csr[0][-1])
op = L.CsrOperator(ctx, *csr, n_cols=n, row_begin=rb)
init = G.start_vector_fast(nl, 1, np.float64, rb)
xd = ctx.to_device(init / np.sqrt(float(np.sum(init * init)) * N)); yd = ctx.empty(nl, np.float64)
rounds = []
for _ in range(3):
    L.spmv(op, xd, yd); ctx.synchronize(); ctx.timer_start()
    for _ in range(20):
        L.spmv(op, xd, yd)
    rounds.append(ctx.timer_stop() / 20)
eng = L.LambdaLanczos(op, n, True, 1)
eng.max_iteration = window; eng.eps = 0.0
eng.init_vector = ctx.to_device(init); eng.eigenvectors_out = ctx.empty((1, nl), np.float64)
eng.run()
ctx.synchronize(); t0 = time.perf_counter(); eng.run(); ctx.synchronize(); wall = time.perf_counter() - t0
ctx.set_profiling(True); eng.run(); st = eng.last_stats
it = eng.getIterationCounts()[0]
print(json.dumps({"ranks": N, "band": band, "rows_per_rank": nl, "nnz_per_rank": nnz_local, "spmv_kernel": op.selected_spmv(),
                  "creation_ms_csr_pb_tiled": [round(op.autotune_ms_of(k), 4) for k in (0, 1, 2)], "tiled_row_blocks_all_own": list(op.tiled_layout()),
                  "spmv_ms_incl_local_copies": sorted(rounds)[1], "iterations": it, "window_wall_ms": wall * 1e3,
                  "per_iteration_us": {"operator_incl_local_copies": st["seconds_spmv"] / it * 1e6,
                                       "gather_copies_of_the_stand_in": st["seconds_comm_gather"] / it * 1e6,
                                       "allreduce_stand_in": st["seconds_comm_allreduce"] / it * 1e6,
                                       "gram_schmidt": st["seconds_orth"] / it * 1e6, "wall": wall / it * 1e6}}))
""" % ROOT


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--child":  # in-process (for rocprofv3: export LL_COMM_PLUGIN first)
        sys.argv = [sys.argv[0], sys.argv[2]]
        exec(compile(CHILD, "<shard child>", "exec"), {"__name__": "__main__"})
        return
    ranks = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]
    for N in ranks:
        env = dict(os.environ, LL_COMM_PLUGIN=SOLO)
        r = subprocess.run([sys.executable, "-c", CHILD, str(N)], capture_output=True, text=True, timeout=900, env=env)
        out = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        print(out[-1] if out else json.dumps({"ranks": N, "error": (r.stderr or r.stdout)[-400:]}), flush=True)


if __name__ == "__main__":
    main()
