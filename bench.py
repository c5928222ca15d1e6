#!/usr/bin/env python3
"""bench.py — Lanczos iterations/s + CSR SpMV GB/s on MI355X (BASELINE.json metric), one process per GPU.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One STEP = one pass of the hot path over the synthetic input: LambdaLanczos::run() on the configured matrix with a
fixed window of `--window` Lanczos iterations (max_iteration = window; the cost of an iteration grows with k, so the
window is part of the metric and is printed in `config`).  value = steps * window / time = Lanczos iterations/s of the
whole job.  With N > 1 the SAME matrix is row-partitioned over the N GPUs (BASELINE config 4) => "scaling": "strong".
Inputs are synthetic (SURVEY 8d generators), resident in HBM before the timed region starts.

Extra objects on the JSON line: `roofline` (the CSR SpMV kernel: algorithmic bytes / HIP-event time on the stream
it is launched on, against the 8 TB/s HBM peak), `roofline_orth` (the Gram-Schmidt kernels of the timed windows) and
`cpu_baseline` (the real reference — or the oracle port — on the host cores of the same machine, bounded sample).
"""
import argparse
import json
import os
import sys
import time

# Host thread pools that keep spinning after their last job (OpenMP workers with an active wait policy, numpy's
# OpenBLAS pool after np.linalg.norm) hit launch-bound timed steps with one ~80 ms stall per process
# (tools/stall_probe.py, profiles/r02_host_thread_stalls.txt): they sleep between jobs / stay single-threaded here.
os.environ.setdefault("OMP_WAIT_POLICY", "passive")
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md); ~6.3 TB/s achievable copy


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c3", choices=["c3", "c3band", "c2", "c5"],
                    help="c3: random symmetric CSR n=1e7 nnz=1.5e8 (metric config); c3band: banded variant; "
                         "c2: 5-pt Laplacian n=1e6; c5: complex torus n=1e6 (Exponentiator)")
    ap.add_argument("--operator", default="csr", choices=["csr", "lattice"],
                    help="c2 / c5: lattice = the matrix-free lattice operator (halo exchange instead of the all-gather)")
    ap.add_argument("--size", dest="n", type=int, default=0, help="override the problem size (grid side for c2/c5)")
    ap.add_argument("--window", type=int, default=100, help="Lanczos iterations per step (max_iteration)")
    ap.add_argument("--spmv-reps", type=int, default=20)
    ap.add_argument("--cpu-window", type=int, default=-1,
                    help="iterations of the CPU baseline sample (default -1: the headline window itself, so that cpu_baseline.value "
                         "is like-for-like with `value` — about 105 s of one host core for config 3 at window 100; 0 = skip).  A "
                         "second, short sample (--cpu-short-window) is reported as cpu_baseline_short")
    ap.add_argument("--cpu-short-window", type=int, default=14,
                    help="iterations of the short CPU sample (cpu_baseline_short; also the window of the all-cores and host-callback legs)")
    ap.add_argument("--eps", type=float, default=None,
                    help="override the engine's eps (0 = never converge: every run does exactly --window iterations)")
    ap.add_argument("--orth-mode", type=int, default=0)
    ap.add_argument("--tridiag-mode", type=int, default=None,
                    help="LL_TRIDIAG_*: 0 QR every iteration, 1 bisection, 2 auto (default: the library's default, auto)")
    ap.add_argument("--host-io", action="store_true",
                    help="hand the start vector / input over and take the results back in HOST buffers (the reference's "
                         "std::vector boundary: PCIe copies inside the timed region); default: device buffers, i.e. inputs "
                         "resident in HBM when the timed region starts")
    ap.add_argument("--no-spmv-variants", action="store_true",
                    help="do not time the other phase-2 forms of the PB SpMV (spmv.ms_by_kernel then holds the selected kernel only)")
    ap.add_argument("--no-phase-timers", action="store_true",
                    help="skip the instrumented steps after the timed region (no per-phase split, no roofline_orth)")
    ap.add_argument("--phase-timers-inline", action="store_true",
                    help="record the per-phase HIP events inside the timed steps instead of in separate steps after them")
    ap.add_argument("--no-exchange-tuning", action="store_true",
                    help="N > 1: skip the pre-pass that times LL_GATHER_CHUNKS x LL_COMM_OVERLAP on the actual shards and keeps the fastest")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the short config-2 / config-5 / host-callback legs that follow the headline run at N = 1")
    ap.add_argument("--tuning", action="append", default=[], metavar="KEY=VALUE",
                    help="per-context setting for this run (ll_ctx_set_tuning; UNSTABLE keys, INTEGRATION.md section 8): A/B "
                         "measurements such as --tuning sweep_pipeline=0; may be given several times; echoed in config.tuning")
    ap.add_argument("--watchdog", type=float, default=1500.0,
                    help="seconds after which a job that has not finished prints a diagnostic and exits with code 3 "
                         "(a hung collective must not look like a slow run); 0 = off")
    ap.add_argument("--dry-run-dist", action="store_true",
                    help="CPU-only check of the multi-process plumbing (rendezvous, id broadcast, partition, shard "
                         "generation, reductions over ranks); no device work, prints one JSON line per job")
    return ap.parse_args()


def spmv_bytes(n, nnz, complex_):
    """Algorithmic bytes of one CSR SpMV: values + int32 columns + int32 row_ptr + x once + y once."""
    s = 16 if complex_ else 8
    return (s + 4) * nnz + 4 * (n + 1) + 2 * s * n


def kernel_sources_sha16():
    """Fingerprint of the kernel sources (what a committed PMC summary was measured on; tools/pmc_summary.py stores it)."""
    import hashlib

    h = hashlib.sha256()
    for name in ("kernels.hip", "spmv_pb.hip", "dev_helpers.hpp", "fixed_round.hpp"):
        with open(os.path.join(ROOT, "lambda-lanczos_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def pmc_traffic(workload, kernels, dtype_tag):
    """HBM bytes per launch from the committed rocprofv3 --pmc summary of this workload (tools/pmc_summary.py), its file
    name and its age, or (None, None, age) when no profile exists or a kernel of this run is not in it (a renamed or
    re-templated kernel: the summary is stale, no number is better than an old one).  bench.py cannot collect PMC counters
    itself."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_%s_pmc_traffic.json" % workload)))
    if not files:
        return None, None, None
    with open(files[-1]) as f:
        d = json.load(f)
    age = {"summary": os.path.relpath(files[-1], ROOT), "collected_at_head": d.get("collected_at_head"),
           "kernel_sources_sha16_then": d.get("kernel_sources_sha16"), "kernel_sources_sha16_now": kernel_sources_sha16()}
    age["kernel_sources_unchanged"] = age["kernel_sources_sha16_then"] == age["kernel_sources_sha16_now"]
    total = 0.0
    for k in kernels:
        # template arguments after the scalar type (pipeline depths, index width) vary: match on name + scalar type
        hits = [v for name, v in d["kernels"].items() if name.startswith("%s<%s" % (k, dtype_tag))]
        if not hits:
            age["kernels_match"] = False
            return None, None, age
        e = max(hits, key=lambda v: v.get("launches", 0))
        total += e["fetch_bytes_mean"] + e["write_bytes_mean"]
    age["kernels_match"] = True
    return total, os.path.relpath(files[-1], ROOT), age


def pmc_orth_traffic(workload, window):
    """Measured HBM bytes of the Gram-Schmidt kernels (lagged sweep + folds, or mdot + maxpy + scale) per window of `window` iterations, from the
    committed PMC summary of the same workload and window (None when there is none)."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_%s_pmc_traffic.json" % workload)))
    if not files:
        return None, None
    with open(files[-1]) as f:
        d = json.load(f)
    if d.get("window_iterations") != window or "orth_bytes_per_window" not in d:
        return None, None
    return d["orth_bytes_per_window"], os.path.relpath(files[-1], ROOT)


STAGE = ["start"]  # where the job is, for the watchdog's diagnostic


def _events_ms(ctx, fn, reps):
    """HIP-event time of `reps` calls of fn on the library stream, per call, median of three rounds."""
    rounds = []
    for _ in range(3):
        fn()
        ctx.synchronize()
        ctx.timer_start()
        for _ in range(reps):
            fn()
        rounds.append(ctx.timer_stop() / reps)
    return sorted(rounds)[1]


def other_config_c2(ctx, L, G):
    """BASELINE config 2 in the same process: 5-point Laplacian n = 1e6, smallest pair, offset -8; two steps of window 100 from
    device buffers, the SpMV kernel by HIP events, and a 24-iteration window against the real reference (or the oracle port)."""
    import oracle_lib

    side, n, window = 1000, 1000 * 1000, 100
    csr = G.laplace2d(side)
    init = G.start_vector_fast(n, 1)
    op = L.CsrOperator(ctx, *csr)
    nnz = int(csr[0][-1])
    xd, yd = ctx.to_device(init / np.linalg.norm(init)), ctx.empty(n, np.float64)
    ms = _events_ms(ctx, lambda: L.spmv(op, xd, yd), 50)
    b = spmv_bytes(n, nnz, False)
    eng = L.LambdaLanczos(op, n, False, 1)
    eng.eigenvalue_offset = -8.0
    eng.max_iteration = window
    eng.init_vector = ctx.to_device(init)
    eng.eigenvectors_out = ctx.empty((1, n), np.float64)
    eng.run()
    ctx.synchronize()
    t0 = time.perf_counter()
    its = 0
    for _ in range(2):
        eng.run()
        its += eng.getIterationCounts()[0]
    ctx.synchronize()
    dt = time.perf_counter() - t0
    lagged = int(eng.last_stats["lagged_iterations"])
    kind = "reference" if oracle_lib.have_reference() else "port"
    chk = oracle_lib.reference() if kind == "reference" else oracle_lib.oracle()
    w = 24
    r = chk.lanczos(csr, init, False, max_iteration=w, offset=-8.0)
    eng.max_iteration = w
    eng.eigenvectors_out = None
    eng.init_vector = lambda v, *_: np.copyto(v, init)
    vals, vecs = eng.run()
    norm = 8.0 + 8.0
    da = float(np.max(np.abs(eng.last_alpha - r["alpha"][:w])))
    db = float(np.max(np.abs(eng.last_beta[: w - 1] - r["beta"][: w - 1])))
    dl = float(abs(vals[0] - r["eigenvalues"][0]))
    ov = float(1.0 - abs(np.vdot(vecs[0], r["eigenvectors"][0])))
    out = {"workload": "5-point Laplacian 1000x1000 fp64, smallest pair, eigenvalue_offset -8", "n": n, "nnz": nnz,
           "value": its / dt, "unit": "Lanczos iterations/s", "steps": 2, "window": window, "ms_per_step": dt / 2 * 1e3,
           "lagged_iterations_last_step": lagged, "io": "device buffers",
           "spmv": {"ms": ms, "algorithmic_bytes": b, "GBps": b / (ms * 1e-3) / 1e9, "frac_of_8TBps": b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "kernel": {L.capi.SPMV_CSR_STREAM: "spmv_stream", L.capi.SPMV_PB: "pb_phase1+pb_phase2_fixed",
                               L.capi.SPMV_TILED: "tl_xmax+tl_spmv"}.get(op.selected_spmv())},
           "cpu_same_window": {"kind": kind, "window": w, "value": w / r["t_total"], "seconds": r["t_total"], "cores": 1},
           "parity_same_window": {"window": w, "max_abs_dalpha": da, "max_abs_dbeta": db, "abs_dlambda": dl,
                                  "eigenvector_one_minus_overlap": ov,
                                  "tolerance": "|dalpha|,|dbeta| <= 1e-10*||A+offset||_inf (16), |dlambda| <= 1e-10*max(1,|lambda+offset|), 1-overlap <= 1e-8",
                                  "ok": bool(da <= 1e-10 * norm and db <= 1e-10 * norm and dl <= 1e-10 * 8.0 and ov <= 1e-8)}}
    op.close()
    return out


def other_config_c5(ctx, L, G):
    """BASELINE config 5 in the same process: complex Hermitian torus n = 1e6, exp(-5iH) v to convergence from device buffers,
    the complex SpMV by HIP events, and the whole run against the real Exponentiator (or the oracle port)."""
    import oracle_lib

    side, n = 1000, 1000 * 1000
    csr = G.torus(side)
    init = G.start_vector_fast(n, 1, np.complex128)
    op = L.CsrOperator(ctx, *csr)
    nnz = int(csr[0][-1])
    xd, yd = ctx.to_device(init / np.linalg.norm(init)), ctx.empty(n, np.complex128)
    ms = _events_ms(ctx, lambda: L.spmv(op, xd, yd), 50)
    b = spmv_bytes(n, nnz, True)
    eng = L.Exponentiator(op, n)
    d_in, d_out = ctx.to_device(init), ctx.empty((n,), np.complex128)
    eng.run(-5.0j, d_in, out=d_out)
    ctx.synchronize()
    t0 = time.perf_counter()
    its = 0
    for _ in range(5):
        its += eng.run(-5.0j, d_in, out=d_out)[1]
    ctx.synchronize()
    dt = time.perf_counter() - t0
    # the same run with full_orthogonalize (EX:63,120-122): Gram-Schmidt against the whole basis in every iteration, two iterations
    # per sweep (pair form) — and with LL_PAIR_GS=0 one sweep per iteration, for the A/B
    full = {}
    eng.full_orthogonalize = True
    for label, key in (("pair_form", None), ("one_sweep_form", "0")):
        ctx.set_tuning("pair_gs", key)
        try:
            eng.run(-5.0j, d_in, out=d_out)
            ctx.synchronize()
            tf = time.perf_counter()
            itf = 0
            for _ in range(5):
                itf += eng.run(-5.0j, d_in, out=d_out)[1]
            ctx.synchronize()
            tf = time.perf_counter() - tf
            full[label] = {"value": itf / tf, "iterations_per_step": itf / 5, "ms_per_step": tf / 5 * 1e3,
                           "pair_iterations_last_step": int(eng.last_stats["pair_iterations"])}
        finally:
            ctx.set_tuning("pair_gs", None)
    o_full, o_full_it, _ = (oracle_lib.reference() if oracle_lib.have_reference() else oracle_lib.oracle()).expo(
        csr, -5.0j, init, full_orthogonalize=True)
    g_full, g_full_it = eng.run(-5.0j, init)
    full["parity"] = {"iterations_cpu": int(o_full_it), "iterations_gpu": int(g_full_it),
                      "max_abs_diff_over_input_norm": float(np.max(np.abs(g_full - o_full)) / np.linalg.norm(init))}
    full["parity"]["ok"] = bool(full["parity"]["max_abs_diff_over_input_norm"] <= 1e-10 and abs(int(o_full_it) - int(g_full_it)) <= 1)
    eng.full_orthogonalize = False
    kind = "reference" if oracle_lib.have_reference() else "port"
    chk = oracle_lib.reference() if kind == "reference" else oracle_lib.oracle()
    o_out, o_it, o_t = chk.expo(csr, -5.0j, init)
    g_out, g_it = eng.run(-5.0j, init)
    err = float(np.max(np.abs(g_out - o_out)) / np.linalg.norm(init))
    ovl = float(1.0 - abs(np.vdot(o_out, g_out)) / (np.linalg.norm(o_out) * np.linalg.norm(g_out)))
    unit = float(abs(np.linalg.norm(g_out) / np.linalg.norm(init) - 1.0))
    out = {"workload": "complex Hermitian torus 1000x1000 (c128), Exponentiator exp(-5iH) v to convergence", "n": n, "nnz": nnz,
           "value": its / dt, "unit": "Lanczos iterations/s", "steps": 5, "iterations_per_step": its / 5, "ms_per_step": dt / 5 * 1e3,
           "io": "device buffers",
           "spmv": {"ms": ms, "algorithmic_bytes": b, "GBps": b / (ms * 1e-3) / 1e9, "frac_of_8TBps": b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "kernel": {L.capi.SPMV_CSR_STREAM: "spmv_stream", L.capi.SPMV_PB: "pb_phase1+pb_phase2_fixed",
                               L.capi.SPMV_TILED: "tl_xmax+tl_spmv"}.get(op.selected_spmv())},
           "full_orthogonalize": full,
           "cpu_whole_run": {"kind": kind, "value": o_it / o_t["t_total"], "seconds": o_t["t_total"], "iterations": int(o_it), "cores": 1},
           "parity_whole_run": {"iterations_cpu": int(o_it), "iterations_gpu": int(g_it), "max_abs_diff_over_input_norm": err,
                                "one_minus_overlap": ovl, "norm_drift": unit,
                                "tolerance": "|out_gpu - out_cpu| <= 1e-10 |in|, 1 - overlap <= 10 eps, | |out|/|in| - 1 | <= 1e-12, iterations within 1",
                                "ok": bool(err <= 1e-10 and ovl <= 10 * 2.3e-16 + 1e-15 and unit <= 1e-12 and abs(int(o_it) - int(g_it)) <= 1)}}
    op.close()
    return out


def callback_leg(ctx, L, csr, n, init, find_max, offset, window, device_value):
    """The unmodified-user-lambda path (LL:126,200-208) on the headline matrix: the same CSR as a HOST mv_mul (scipy's CSR row
    loop), `window` iterations.  Every iteration moves one n-vector down and one up over PCIe and runs the user's code on
    the host; the split below says where a user of the plain drop-in spends the time."""
    import scipy.sparse as sp

    A = sp.csr_matrix((csr[2], csr[1], csr[0]), shape=(n, n))
    t_cb = [0.0, 0]

    def mv_mul(x, out):
        t = time.perf_counter()
        out += A @ x
        t_cb[0] += time.perf_counter() - t
        t_cb[1] += 1

    op = L.HostOperator(ctx, mv_mul, n)
    eng = L.LambdaLanczos(op, n, find_max, 1)
    eng.eigenvalue_offset = offset
    eng.max_iteration = window
    eng.init_vector = lambda v, *_: np.copyto(v, init)
    ctx.set_profiling(True)
    t0 = time.perf_counter()
    vals, _ = eng.run()
    dt = time.perf_counter() - t0
    ctx.set_profiling(False)
    st = eng.last_stats
    its = eng.getIterationCounts()[0]
    op.close()
    copies = max(st["seconds_spmv"] - t_cb[0], 0.0)
    return {"value_callback": its / dt, "unit": "Lanczos iterations/s", "window": window, "iterations": its,
            "callback_calls": t_cb[1], "eigenvalue": float(vals[0]),
            "per_iteration_ms": {"user_mv_mul_on_host": t_cb[0] / its * 1e3,
                                 "d2h_plus_h2d_of_one_vector_each_and_offset_dot": copies / its * 1e3,
                                 "gram_schmidt_on_device": st["seconds_orth"] / its * 1e3,
                                 "everything": dt / its * 1e3},
            "bytes_over_pcie_per_iteration": 2 * 8 * n,
            "device_operator_same_window_value": device_value,
            "note": "mv_mul = scipy CSR row loop on one host core; the device-resident operator runs the same window at "
                    "device_operator_same_window_value"}


def start_watchdog(seconds, rank):
    """A hung RCCL collective (or a rank that never arrives) would otherwise sit silently until the driver's own limit."""
    if seconds <= 0:
        return
    import threading

    def bark():
        sys.stderr.write("bench.py watchdog: rank %d still in stage '%s' after %.0f s - giving up\n" % (rank, STAGE[0], seconds))
        sys.stderr.flush()
        os._exit(3)

    t = threading.Timer(seconds, bark)
    t.daemon = True
    t.start()


def main():
    args = parse_args()
    world = args.gpus
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    start_watchdog(args.watchdog, rank)
    dist = None
    if world > 1:
        # torch.distributed.run pins OMP_NUM_THREADS=1 per worker; the host-side matrix generation and the PB image build
        # (one-off, outside the timed region) are OpenMP loops: give every rank its share of the host cores instead
        os.environ["OMP_NUM_THREADS"] = str(max(1, (os.cpu_count() or 1) // world))
        import torch.distributed as dist  # control plane only (gloo); the data plane is RCCL inside the library

        assert int(os.environ.get("WORLD_SIZE", "1")) == world, "launch with torch.distributed.run --nproc-per-node N"
        dist.init_process_group("gloo")

    import lambda_lanczos_amd as L
    from lambda_lanczos_amd import generators as G

    if args.dry_run_dist:
        class _DryCtx:  # partition() only; everything below the shard generation is skipped
            def partition(self, n_):
                return L.partition(n_, world, rank)

        ctx = _DryCtx()
        box = [os.urandom(128) if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(box, src=0)
        uid = box[0]
    else:
        # LL_BENCH_DEVICE: testing hook — several ranks on one GPU with the host-staged test transport (LL_COMM_PLUGIN)
        ctx = L.Context(int(os.environ.get("LL_BENCH_DEVICE", local_rank)))
        if world > 1:
            STAGE[0] = "communicator init + self-check (all-gather of rank tags, all-reduce of ones)"
            box = [L.Context.unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            ctx.init_comm(box[0], rank, world)   # fails loudly unless all `world` ranks answer in rank order
    for kv in args.tuning:
        if not args.dry_run_dist:
            ctx.set_tuning(*kv.split("=", 1))
    ranks_seen = 1 if args.dry_run_dist else ctx.ranks_seen()
    # which transport answers the collectives: "rccl" | "plugin:<path>" (LL_COMM_PLUGIN: the host-staged test transport) | "none"
    transport = "none" if args.dry_run_dist else ctx.transport()

    # ------------------------------------------------------------ synthetic input, resident in HBM
    STAGE[0] = "matrix generation + upload"
    t_gen = time.time()
    wl = args.workload
    complex_ = wl == "c5"
    if wl in ("c3", "c3band"):
        n = args.n or 10_000_000
        rb, nl = ctx.partition(n)
        band = 65536 if wl == "c3band" else 0
        if band and n < 4 * band:
            band = max(2, n // 8)
        csr = G.randsym(n, band=band, row_begin=rb, n_local=nl)
        find_max, offset = True, 0.0
        name = "random symmetric CSR n=%d nnz=%d fp64%s" % (n, 15 * n, " (banded +-%d)" % band if band else "")
    elif wl == "c2":
        side = args.n or 1000
        n = side * side
        rb, nl = ctx.partition(n)
        csr = G.laplace2d(side, rb, nl)
        find_max, offset = False, -8.0
        name = "5-point Laplacian %dx%d fp64" % (side, side)
    else:
        side = args.n or 1000
        n = side * side
        rb, nl = ctx.partition(n)
        csr = G.torus(side, rb, nl)
        find_max, offset = False, 0.0
        name = "complex Hermitian torus %dx%d" % (side, side)
    dtype = np.complex128 if complex_ else np.float64
    nnz_local = int(csr[0][-1])
    nnz = nnz_local
    if world > 1:
        import torch

        t = torch.tensor([nnz_local], dtype=torch.int64)
        dist.all_reduce(t)
        nnz = int(t.item())
    init = G.start_vector_fast(nl, 1, dtype, rb)
    if args.dry_run_dist:
        import hashlib

        import torch

        t = torch.tensor([float(nl), float(np.sum(init.real))], dtype=torch.float64)
        if dist is not None:
            dist.all_reduce(t)
            dist.barrier()
        if rank == 0:
            print(json.dumps({"dry_run": True, "world": world, "n": n, "nnz": nnz, "rows_total": int(t[0].item()),
                              "start_vector_sum": float(t[1].item()), "id_sha": hashlib.sha1(uid).hexdigest()}), flush=True)
        if dist is not None:
            dist.destroy_process_group()
        return
    t_up = 0.0
    lattice = args.operator == "lattice" and wl in ("c2", "c5")
    if lattice and wl == "c2":
        op = L.StencilOperator(ctx, [side, side], diag=4.0, hop=-1.0, row_begin=rb, n_local=nl)
        name += ", matrix-free lattice operator"
    elif lattice:  # config 5: Landau-gauge Peierls phases on the x hops, random on-site terms (generators.torus_np)
        import math

        onsite = G.u01(np.arange(rb, rb + nl, dtype=np.uint64)) - 0.5
        op = L.StencilOperator(ctx, [side, side], diag=0.0, hop=[-1.0, -1.0], periodic=True, onsite=onsite,
                               dtype=np.complex128, row_begin=rb, n_local=nl,
                               phase_grad=[[0.0, 0.0], [2.0 * math.pi * 3.0 / side, 0.0]])
        name += ", matrix-free lattice operator"
    else:
        t_up = time.time()
        op = L.CsrOperator(ctx, *csr, n_cols=n, row_begin=rb)
        t_up = time.time() - t_up
    t_gen = time.time() - t_gen

    def barrier():
        ctx.synchronize()
        if dist is not None:
            dist.barrier()

    def max_over_ranks(v):
        if dist is None:
            return v
        import torch

        t = torch.tensor([v], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # ------------------------------------------------------------ SpMV kernel: HIP events on its own stream
    STAGE[0] = "SpMV timing"
    xd = ctx.to_device(init / np.linalg.norm(init))
    yd = ctx.empty(nl, dtype)
    # lattice: x read once, y written once (+ the real on-site array of config 5)
    b_spmv = (2 * (16 if complex_ else 8) * n + (8 * n if complex_ else 0)) if lattice else spmv_bytes(n, nnz, complex_)
    selected = -1 if lattice else op.selected_spmv()
    # phase 2 of the PB SpMV: pb_phase2_fixed (order-independent fixed-point sums, the default) or pb_phase2
    # (LL_PB_PHASE2=ordered|atomic)
    phase2_form = os.environ.get("LL_PB_PHASE2", "fixed")
    if phase2_form not in ("ordered", "atomic"):
        phase2_form = "fixed"
    p2 = "pb_phase2_fixed" if phase2_form == "fixed" else "pb_phase2"
    kernel_names = {L.capi.SPMV_CSR_STREAM: "spmv_stream", L.capi.SPMV_PB: "pb_phase1+" + p2, L.capi.SPMV_TILED: "tl_xmax+tl_spmv",
                    -1: "stencil_kernel"}
    # The operator timed both kernels on the actual matrix when it was created and released the slower image; those
    # creation-time figures are reported next to the event timing of the kernel that is in use.
    tune = None if lattice else {kernel_names[k]: op.autotune_ms_of(k) for k in
                                 (L.capi.SPMV_CSR_STREAM, L.capi.SPMV_PB, L.capi.SPMV_TILED)}
    def time_spmv(o):
        rounds = []
        for rnd in range(3):
            L.spmv(o, xd, yd)
            barrier()
            ctx.timer_start()
            for _ in range(args.spmv_reps):
                L.spmv(o, xd, yd)
            rounds.append(max_over_ranks(ctx.timer_stop() / args.spmv_reps))
        return sorted(rounds)[len(rounds) // 2]

    # ------------------------------------------------------------ N > 1: how the exchange is cut and issued, timed on the actual shards
    # LL_GATHER_CHUNKS in {1, 2, 4} (pieces of the all-gather: fixed in the PB image's column-block table, so one operator per
    # value) x LL_COMM_OVERLAP in {0, 1} (own-column work under the gather or everything on one stream: read per launch), three
    # SpMVs each behind one warm-up, max over ranks; the fastest combination is what the timed steps run.  Every rank sees the same
    # maxima and takes the same pick.
    exchange_tuning = None
    if world > 1 and selected == L.capi.SPMV_PB and not args.no_exchange_tuning:
        STAGE[0] = "exchange tuning pre-pass"
        table, ops_by_chunks = {}, {}
        ctx.set_tuning("spmv_kernel", "pb")
        for chunks in (1, 2, 4):
            ctx.set_tuning("gather_chunks", chunks)
            cand = L.CsrOperator(ctx, *csr, n_cols=n, row_begin=rb)
            ops_by_chunks[chunks] = cand
            for overlap in (1, 0):
                ctx.set_tuning("comm_overlap", overlap)
                L.spmv(cand, xd, yd)
                barrier()
                ctx.timer_start()
                for _ in range(3):
                    L.spmv(cand, xd, yd)
                table["chunks=%d,overlap=%d" % (chunks, overlap)] = max_over_ranks(ctx.timer_stop() / 3)
        best = min(table, key=lambda k: (table[k], k))
        pick_chunks, pick_overlap = int(best.split(",")[0].split("=")[1]), int(best.split(",")[1].split("=")[1])
        ctx.set_tuning("gather_chunks", pick_chunks)
        ctx.set_tuning("comm_overlap", pick_overlap)
        ctx.set_tuning("spmv_kernel", None)
        for chunks, cand in ops_by_chunks.items():
            if chunks != pick_chunks:
                cand.close()
        op.close()
        op = ops_by_chunks[pick_chunks]
        exchange_tuning = {"ms_per_spmv_max_over_ranks": table, "pick": {"LL_GATHER_CHUNKS": pick_chunks, "LL_COMM_OVERLAP": pick_overlap},
                           "how": "3 SpMVs per combination behind one warm-up, HIP events on the library stream, max over ranks; "
                                  "the operator of the timed steps is the picked one"}

    spmv_ms = time_spmv(op)
    spmv_variants = {"%s [%s]" % (kernel_names[selected], phase2_form) if selected == L.capi.SPMV_PB else kernel_names[selected]: spmv_ms}
    spmv_gbs = b_spmv / (spmv_ms * 1e-3) / 1e9
    if selected == L.capi.SPMV_PB and world == 1 and not args.no_spmv_variants:
        # The other two phase-2 forms on the SAME matrix in the SAME process (each its own image, built and released here),
        # so that a slow box and a slow kernel can be told apart from one JSON line.
        STAGE[0] = "SpMV timing of the other phase-2 forms"
        saved = {k: os.environ.get(k) for k in ("LL_PB_PHASE2", "LL_SPMV_KERNEL")}
        try:
            for form in ("fixed", "ordered", "atomic"):
                if form == phase2_form:
                    continue
                os.environ["LL_PB_PHASE2"], os.environ["LL_SPMV_KERNEL"] = form, "pb"
                ctx.reload_env()
                alt = L.CsrOperator(ctx, *csr, n_cols=n, row_begin=rb)
                nm = "pb_phase1+%s [%s]" % ("pb_phase2" if form != "fixed" else "pb_phase2_fixed", form)
                spmv_variants[nm] = time_spmv(alt)
                alt.close()
        finally:
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
            ctx.reload_env()
        spmv_variants["%s [%s] again, after the others" % (kernel_names[selected], phase2_form)] = time_spmv(op)

    # ------------------------------------------------------------ measured streaming ceilings of THIS device, same process (SURVEY 8d "Bound")
    STAGE[0] = "bandwidth ceilings"
    probe_bytes = 2 << 30
    read_ceiling, copy_ceiling = ctx.bandwidth_probe(probe_bytes)
    ceiling = {"read_GBps": read_ceiling, "copy_GBps": copy_ceiling, "bytes": probe_bytes,
               "how": "ll_bandwidth_probe: a read-only sum and a copy kernel over 2 GiB, 16-byte accesses, best grid of four, three "
                      "launches each by HIP events on the library stream, in this process right after the SpMV timing; copy = bytes read + "
                      "bytes written per second"}

    # ------------------------------------------------------------ timed steps
    STAGE[0] = "timed Lanczos windows"
    itern = []
    stats_acc = {"lagged_iterations": 0, "seconds_spmv": 0.0, "seconds_orth": 0.0, "seconds_host_tridiag": 0.0, "seconds_host_enqueue": 0.0,
                 "seconds_host_wait": 0.0, "seconds_setup": 0.0, "seconds_finish": 0.0, "seconds_total": 0.0,
                 "seconds_comm_gather": 0.0, "seconds_comm_allreduce": 0.0}

    if wl == "c5":
        eng = L.Exponentiator(op, n)
        eng.max_iteration = args.window

        if args.host_io:
            def step():
                out, it = eng.run(-1j * 5.0, init)
                return it
        else:
            d_in, d_out = ctx.to_device(init), ctx.empty((nl,), dtype)

            def step():
                out, it = eng.run(-1j * 5.0, d_in, out=d_out)
                return it
    else:
        eng = L.LambdaLanczos(op, n, find_max, 1)
        eng.max_iteration = args.window
        eng.eigenvalue_offset = offset
        eng.orth_mode = args.orth_mode
        if args.tridiag_mode is not None:
            eng.tridiag_mode = args.tridiag_mode
        if args.eps is not None:
            eng.eps = args.eps
        if args.host_io:
            eng.init_vector = lambda v, *_: np.copyto(v, init)
        else:
            eng.init_vector = ctx.to_device(init)
            eng.eigenvectors_out = ctx.empty((1, nl), dtype)

        def step():
            eng.run()
            return eng.getIterationCounts()[0]

    # The timed steps run WITHOUT the library's per-phase HIP events (three event records per iteration cost 5 % at
    # n = 1e6 and 20 % on the launch-bound config 5); the per-phase split comes from the same number of steps (at most
    # three) repeated WITH them right after the timed region.  --phase-timers-inline restores the old single pass.
    inline = args.phase_timers_inline and not args.no_phase_timers
    ctx.set_profiling(inline)
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    lagged_timed = 0
    pair_timed = 0
    for _ in range(args.steps):
        itern.append(step())
        lagged_timed += int((eng.last_stats or {}).get("lagged_iterations", 0))
        pair_timed += int((eng.last_stats or {}).get("pair_iterations", 0))
        if inline:
            for key in stats_acc:
                stats_acc[key] += eng.last_stats[key]
    barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0)
    total_iters = int(sum(itern))
    value = total_iters / elapsed
    itern_phases = list(itern)
    # The same steps through the OTHER boundary, in the same process: `value` is the device-buffer figure unless --host-io
    # was given; value_host_io is always the reference's std::vector boundary (LL:330: start vector up through the pinned
    # staging buffer, eigenvector / output back), value_device_io always the HBM-resident one.
    STAGE[0] = "timed windows through the other I/O boundary"
    if wl == "c5":
        if args.host_io:
            d_in2, d_out2 = ctx.to_device(init), ctx.empty((nl,), dtype)

            def step_other():
                return eng.run(-1j * 5.0, d_in2, out=d_out2)[1]
        else:
            def step_other():
                return eng.run(-1j * 5.0, init)[1]
    else:
        keep_io = (eng.init_vector, getattr(eng, "eigenvectors_out", None))
        other_io = ((ctx.to_device(init), ctx.empty((1, nl), dtype)) if args.host_io else
                    ((lambda v, *_: np.copyto(v, init)), None))

        def step_other():
            eng.init_vector, eng.eigenvectors_out = other_io
            try:
                eng.run()
            finally:
                eng.init_vector, eng.eigenvectors_out = keep_io
            return eng.getIterationCounts()[0]
    step_other()
    barrier()
    t0o = time.perf_counter()
    iters_other = sum(step_other() for _ in range(args.steps))
    barrier()
    elapsed_other = max_over_ranks(time.perf_counter() - t0o)
    value_other = iters_other / elapsed_other
    value_host_io, value_device_io = (value, value_other) if args.host_io else (value_other, value)
    if not inline and not args.no_phase_timers:
        STAGE[0] = "instrumented steps (per-phase timers)"
        ctx.set_profiling(True)
        itern_phases = []
        for _ in range(min(args.steps, 3)):
            itern_phases.append(step())
            for key in stats_acc:
                stats_acc[key] += eng.last_stats[key]
        barrier()

    # Gram-Schmidt kernels of the instrumented windows: algorithmic bytes (minimal-pass model minus the SpMV) / device time
    s = 16 if complex_ else 8
    two_sweep_bytes = sum(s * n * (2 * k + 9) for it in itern_phases for k in range(1, it + 1))
    lagged_gs = lagged_timed > 0   # the Gram-Schmidt form of the TIMED steps (ll_run_stats.lagged_iterations)
    if wl == "c5":
        orth_bytes = sum(s * n * 9 for it in itern_phases for _k in range(1, it + 1))
        orth_model = "s*n*9 bytes per iteration (no Gram-Schmidt against the basis: Exponentiator default)"
    elif lagged_gs and pair_timed > 0:
        # pair form: iterations 1 and 2 single (s*n*(k+4)), then pairs (k, k+1): the first three-term kernel (3R 1W) + ONE sweep
        # that reads the k-2 stored vectors and the four raw ones and writes u_{k-2}, u_{k-1} and the compensated r4
        def pair_window(it):
            tot, k = 0, 1
            while k <= it:
                if k >= 3:
                    tot += s * n * (k + 9)
                    k += 2
                else:
                    tot += s * n * (k + 4)
                    k += 1
            return tot
        orth_bytes = sum(pair_window(it) for it in itern_phases)
        orth_model = ("TWO iterations per sweep (pair form, %d of the %d timed iterations): s*n*(k+9) bytes per pair (k, k+1) = "
                      "three-term kernel (3R 1W) + one sweep over k-2 stored and 4 raw vectors (3 written); the one-sweep form needs "
                      "s*n*(k+4) per iteration, the two-sweep form of SURVEY 8d s*n*(2k+9)" % (pair_timed, total_iters))
    elif lagged_gs:
        # one-sweep (lagged) form: reads y, r, u_{k-2} and the k-1 complete basis vectors, writes w and u_{k-1}
        orth_bytes = sum(s * n * (k + 4) for it in itern_phases for k in range(1, it + 1))
        orth_model = ("s*n*(k+4) bytes per iteration: the one-sweep (lagged) Gram-Schmidt form this run used; the two-sweep "
                      "form of SURVEY 8d needs s*n*(2k+9)")
    else:
        orth_bytes = two_sweep_bytes
        orth_model = "s*n*(2k+9) bytes per iteration (SURVEY 8d minimal-pass model)"
    orth_s = max_over_ranks(stats_acc["seconds_orth"])
    spmv_loop_s = max_over_ranks(stats_acc["seconds_spmv"])
    comm_gather_s = max_over_ranks(stats_acc["seconds_comm_gather"])
    comm_allreduce_s = max_over_ranks(stats_acc["seconds_comm_allreduce"])
    setup_by_rank = [t_gen]
    upload_by_rank = [t_up]
    gather_by_rank = [stats_acc["seconds_comm_gather"]]
    if dist is not None:
        import torch

        tt = [torch.zeros(3, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(tt, torch.tensor([t_gen, t_up, stats_acc["seconds_comm_gather"]], dtype=torch.float64))
        setup_by_rank = [float(t[0]) for t in tt]
        upload_by_rank = [float(t[1]) for t in tt]
        gather_by_rank = [float(t[2]) for t in tt]

    # ------------------------------------------------------------ CPU baseline (rank 0, N = 1 only; bounded sample)
    STAGE[0] = "CPU baseline"
    cpu = None
    if world == 1 and args.cpu_window != 0 and wl == "c5":
        # config 5 runs fully on the CPU in seconds: the real Exponentiator<T>::run (EX:87-173) through
        # oracle/_ref, same matrix, same input, same a = -5i; the whole run is the sample
        import oracle_lib

        kind = "reference" if oracle_lib.have_reference() else "port"
        chk = oracle_lib.reference() if kind == "reference" else oracle_lib.oracle()
        o_out, o_it, o_t = chk.expo((csr[0], csr[1], csr[2]), -5.0j, init, max_iteration=args.window)
        barrier()
        tg = time.perf_counter()
        g_out, g_it = eng.run(-1j * 5.0, init)
        barrier()
        tg = time.perf_counter() - tg
        err = float(np.max(np.abs(g_out - o_out)) / np.linalg.norm(init))
        ovl = float(abs(np.vdot(o_out, g_out)) / (np.linalg.norm(o_out) * np.linalg.norm(g_out)))
        cpu = {
            "value": o_it / o_t["t_total"],
            "unit": "Lanczos iterations/s",
            "cores": 1,
            "kind": kind,
            "sample": "the whole run: Exponentiator::run, exp(-5iH)v on the same matrix and input, %d iterations, "
                      "single thread like the reference" % o_it,
            "seconds": o_t["t_total"],
            "spmv_GBps": b_spmv * o_it / max(o_t["t_mv"], 1e-12) / 1e9,
            "gpu_same_window_value": g_it / tg,
            "parity_same_window": {"iterations_cpu": int(o_it), "iterations_gpu": int(g_it),
                                   "max_abs_diff_over_input_norm": err, "one_minus_overlap": 1.0 - ovl,
                                   "tolerance": "|out_gpu - out_cpu| <= 1e-10 |in|, 1 - overlap <= 10 eps, iterations within 1",
                                   "ok": bool(err <= 1e-10 and 1.0 - ovl <= 10 * 2.3e-16 + 1e-15 and abs(int(o_it) - int(g_it)) <= 1)},
            "host_cores_available": os.cpu_count(),
        }
    cpu_short = None
    if world == 1 and args.cpu_window != 0 and wl != "c5":
        import oracle_lib

        kind = "reference" if oracle_lib.have_reference() else "port"
        chk = oracle_lib.reference() if kind == "reference" else oracle_lib.oracle()
        full = (csr[0], csr[1], csr[2])

        def cpu_leg(w):
            """LambdaLanczos::run of the real reference (or the oracle port) with max_iteration = w on the host, then the same window
            on the GPU: like-for-like rate and eigenpair parity."""
            r = chk.lanczos(full, init, find_max, max_iteration=w, offset=offset, trace=False)
            cpu_its = r["iter_counts"][0]
            eng.max_iteration = w
            step()
            barrier()
            tg = time.perf_counter()
            vals_g, vecs_g = eng.run()
            barrier()
            tg = time.perf_counter() - tg
            d_lam = float(abs(vals_g[0] - r["eigenvalues"][0]))
            vecs_g = vecs_g if isinstance(vecs_g, np.ndarray) else vecs_g.get()
            defect = float(1.0 - abs(np.vdot(r["eigenvectors"][0], vecs_g[0])))
            return {
                "value": cpu_its / r["t_total"],
                "unit": "Lanczos iterations/s",
                "cores": 1,
                "kind": kind,
                "sample": "same matrix and start vector, LambdaLanczos::run with max_iteration=%d (mean k=%.1f), "
                          "single thread like the reference" % (w, (w + 1) / 2),
                "seconds": r["t_total"],
                "note": ("the cost of an iteration grows with k: this sample IS the headline window, `value` and "
                         "`gpu_same_window_value` are like-for-like with the line's `value`" if w == args.window else
                         "the cost of an iteration grows with k: compare `value` (window of %d iterations) with "
                         "`gpu_same_window_value`, NOT with the headline value (window of %d)" % (w, args.window)),
                "spmv_GBps": b_spmv * cpu_its / max(r["t_mv"], 1e-12) / 1e9,
                "gpu_same_window_value": cpu_its / tg,
                "parity_same_window": {"eigenvalue_cpu": float(r["eigenvalues"][0]), "eigenvalue_gpu": float(vals_g[0]),
                                       "abs_diff": d_lam, "eigenvector_one_minus_overlap": defect,
                                       "tolerance": "|dlambda| <= 1e-10*max(1,|lambda|), 1-|<v_cpu,v_gpu>| <= 1e-8",
                                       "ok": bool(d_lam <= 1e-10 * max(1.0, abs(vals_g[0])) and defect <= 1e-8)},
                "host_cores_available": os.cpu_count(),
            }

        w_main = args.window if args.cpu_window < 0 else args.cpu_window
        if args.cpu_short_window > 0 and args.cpu_short_window != w_main:
            STAGE[0] = "CPU baseline (short sample)"
            cpu_short = cpu_leg(args.cpu_short_window)
        STAGE[0] = "CPU baseline (headline window)"
        cpu = cpu_leg(w_main)
        eng.max_iteration = args.window
    short_w = args.cpu_short_window if args.cpu_short_window > 0 else 14

    traffic, traffic_src, traffic_age = (None, None, None)
    orth_traffic, orth_traffic_src = (None, None)
    if world == 1 and not args.n and not lattice:
        traffic, traffic_src, traffic_age = pmc_traffic(wl, kernel_names[selected].split("+"), "ll::zc" if complex_ else "double")
        if args.eps is None and wl != "c5":
            orth_traffic, orth_traffic_src = pmc_orth_traffic(wl, args.window)

    cpu_all = None
    if cpu is not None and wl != "c5":
        # courtesy upper bound: the oracle port with OpenMP over all host cores, same sample
        import oracle_lib

        orc = oracle_lib.oracle()
        best = None
        try:
            for want in (16, 32, 64):  # more threads than ~32 got slower on the GPU box's host (measured)
                if want > (os.cpu_count() or 1):
                    break
                threads = orc.set_threads(want)
                r2 = orc.lanczos((csr[0], csr[1], csr[2]), init, find_max, max_iteration=short_w, offset=offset,
                                 trace=False)
                if best is None or r2["t_total"] < best[1]["t_total"]:
                    best = (threads, r2)
        finally:
            orc.set_threads(1)
        if best is not None:
            threads, r2 = best
            cpu_all = {"value": r2["iter_counts"][0] / r2["t_total"], "unit": "Lanczos iterations/s", "cores": threads,
                       "kind": "port", "seconds": r2["t_total"],
                       "sample": (cpu_short or cpu)["sample"].replace("single thread like the reference",
                                                                      "OpenMP threads (best of 16/32/64; not how the reference runs)"),
                       "spmv_GBps": b_spmv * r2["iter_counts"][0] / max(r2["t_mv"], 1e-12) / 1e9}

    # ------------------------------------------------------------ the other single-GPU configs and the callback path
    other, cb_leg = None, None
    if world == 1 and wl == "c3" and not args.n and not args.no_other_configs:
        STAGE[0] = "host-callback leg"
        try:
            cb_leg = callback_leg(ctx, L, csr, n, init, find_max, offset, max(short_w, 8),
                                  (cpu_short or cpu)["gpu_same_window_value"] if (cpu_short or cpu) else None)
        except Exception as e:  # noqa: BLE001 - reported, never fatal for the headline line
            cb_leg = {"error": repr(e)}
        other = {}
        for key, fn in (("c2", other_config_c2), ("c5", other_config_c5)):
            STAGE[0] = "other config " + key
            t_leg = time.perf_counter()
            try:
                other[key] = fn(ctx, L, G)
            except Exception as e:  # noqa: BLE001
                other[key] = {"error": repr(e)}
            other[key]["leg_wall_s"] = time.perf_counter() - t_leg

    if rank == 0:
        line = {
            # BASELINE.json's metric string for its own configuration; a descriptive one for the other workloads
            "metric": ("Lanczos iterations/sec + SpMV GB/s (fp64, n=10M nnz=150M) at 1/2/4/8 GPUs"
                       if wl == "c3" and not args.n else
                       "Lanczos iterations/sec (fixed window) + SpMV GB/s, %s" % ("complex fp64" if complex_ else "fp64")),
            "value": value,
            "value_host_io": value_host_io,      # the same steps with std::vector-style host buffers at the boundary (LL:330)
            "value_device_io": value_device_io,  # ... with the start vector / result resident in HBM (what `value` is by default)
            "unit": "Lanczos iterations/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "c128" if complex_ else "f64",
            "data": "synthetic",
            "config": {
                "workload": name,
                "n": n,
                "nnz": nnz,
                "step": "one %s::run with max_iteration=%d; %.0f iterations per run (k = 1..%.0f, mean k = %.1f), nroot=%s"
                        % ("Exponentiator" if wl == "c5" else "LambdaLanczos", args.window, total_iters / max(args.steps, 1),
                           total_iters / max(args.steps, 1), (total_iters / max(args.steps, 1) + 1) / 2,
                           "n/a" if wl == "c5" else "5"),
                "iterations_per_step": total_iters / max(args.steps, 1),
                "partition": "single GPU" if world == 1 else "1-D row partition over %d GPUs, RCCL %s" % (
                    world, "halo exchange" if lattice else "all-gather of x in chunks on a second stream, own-column SpMV under it"),
                "orth_mode": args.orth_mode,
                "gram_schmidt": ("no re-orthogonalisation (Exponentiator default, EX:120)" if wl == "c5" else
                                 ("full re-orthogonalisation against all previous Lanczos vectors in every iteration, "
                                  "block classical Gram-Schmidt; " +
                                  (("ONE sweep over the basis per TWO iterations (pair form, DESIGN.md 3.2: %d of the %d timed "
                                    "iterations; LL_PAIR_GS=0 runs one sweep per iteration, LL_FUSE_LAUNCHES=1 two)" % (pair_timed, total_iters))
                                   if pair_timed > 0 else
                                   ("ONE sweep over the basis per iteration: the update is applied one iteration late and its "
                                    "effect on the recurrence is compensated exactly (DESIGN.md 3.2; %d of the %d timed "
                                    "iterations; LL_FUSE_LAUNCHES=1 runs the two-sweep form)" % (lagged_timed, total_iters))
                                   if lagged_gs else "two sweeps over the basis per iteration (multi-dot, multi-axpy)"))),
                "tridiag_mode": int(eng.tridiag_mode) if hasattr(eng, "tridiag_mode") else None,
                "eps": "engine default" if args.eps is None else args.eps,
                "tuning": args.tuning or None,
                "io": ("host buffers at the boundary (PCIe copies inside the timed region)" if args.host_io else
                       "start vector / input and eigenvector / output in device buffers (resident in HBM before the timed "
                       "region; --host-io times the std::vector boundary instead)"),
            },
            # ranks that answered the communicator self-check — counted as RCCL's only when RCCL is the transport that ran
            "rccl_ranks_seen": ranks_seen if (transport == "rccl" or world == 1) else 0,
            "comm_ranks_seen": ranks_seen,
            "transport": transport,
            "exchange_tuning": exchange_tuning,
            "spmv": {"GBps": spmv_gbs, "ms": spmv_ms, "algorithmic_bytes": b_spmv, "frac_of_8TBps": spmv_gbs / HBM_PEAK_GBS,
                     "kernel": kernel_names[selected] + " (picked by timing the candidates at upload; the other images are released)",
                     "ms_by_kernel": spmv_variants, "creation_time_autotune_ms": tune,
                     "includes_exchange": world > 1},
            "roofline": {
                "kernel": kernel_names[selected],
                "launch": "one ll_spmv call y = A x (pb: two back-to-back kernels, phase 2 with order-independent fixed-point "
                          "sums), HIP events on the library stream, %d launches averaged" % args.spmv_reps,
                "bound": "hbm",
                "achieved": spmv_gbs,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": spmv_gbs / HBM_PEAK_GBS,
                "measured_ceiling": ceiling,
                "frac_of_measured": spmv_gbs / read_ceiling,        # against the read-only stream (96 % of the algorithmic bytes are reads)
                "frac_of_measured_copy": spmv_gbs / copy_ceiling,   # against the copy stream (read + write mix)
                "traffic": traffic,
                "traffic_source": traffic_src,
                "traffic_age": traffic_age,
                "achieved_actual_traffic_GBps": (traffic / (spmv_ms * 1e-3) / 1e9) if traffic else None,
            },
            "roofline_orth": {
                "kernel": ("pair_three_term + pair_sweep + folds (two iterations per sweep)" if (lagged_gs and pair_timed > 0) else
                           "lagged sweep + folds (three-term, one-sweep Gram-Schmidt, norm)" if lagged_gs else
                           "mdot+maxpy+scale (three-term, Gram-Schmidt, norm, normalise)"),
                "bound": "hbm",
                "achieved": (orth_bytes / orth_s / 1e9) if orth_s > 0 else None,  # None: --no-phase-timers
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": (orth_bytes / orth_s / 1e9 / HBM_PEAK_GBS) if orth_s > 0 else None,
                "frac_of_measured": (orth_bytes / orth_s / 1e9 / read_ceiling) if orth_s > 0 else None,
                "traffic": orth_traffic,
                "traffic_source": orth_traffic_src,
                "traffic_note": "measured HBM bytes of the Gram-Schmidt kernels (one-sweep form: lagged sweep + folds; two-sweep "
                                "form: mdot + maxpy + scale) per step (one window), PMC counters; "
                                "achieved = algorithmic bytes of the timed steps / their device time",
                "algorithmic_bytes_per_step": orth_bytes / max(len(itern_phases), 1),
                "model": orth_model,
                "lagged_iterations_timed_steps": lagged_timed,
                "pair_iterations_timed_steps": pair_timed,
                "two_sweep_model_bytes_per_step": (two_sweep_bytes / max(len(itern_phases), 1)) if wl != "c5" else None,
            },
            "phases": {
                "steps": len(itern_phases) if not args.no_phase_timers else 0,
                "source": ("the timed steps themselves" if inline else
                           "the same step repeated with per-phase HIP events after the timed region"),
                "device_s_operator": spmv_loop_s,
                "device_s_orth": orth_s,
                "host_s_tridiag": stats_acc["seconds_host_tridiag"],
                "host_s_enqueue": stats_acc["seconds_host_enqueue"],
                "host_s_wait_scalars": stats_acc["seconds_host_wait"],
                "host_s_setup": stats_acc["seconds_setup"],
                "host_s_ritz_finish": stats_acc["seconds_finish"],
                "library_s_total": stats_acc["seconds_total"],
                "wall_s": elapsed,
                "device_s_comm_gather": comm_gather_s,
                "device_s_comm_gather_min_over_ranks": min(gather_by_rank),
                "device_s_comm_gather_max_over_ranks": max(gather_by_rank),
                "device_s_comm_gather_by_rank": gather_by_rank,
                "device_s_comm_allreduce": comm_allreduce_s,
                "setup_s_generate_upload": max(setup_by_rank),
                "setup_s_generate_upload_by_rank": setup_by_rank,
                "setup_s_operator_create_by_rank": upload_by_rank,
            },
            "cpu_baseline": cpu,
            "cpu_baseline_short": cpu_short,
            "cpu_baseline_all_cores": cpu_all,
            "callback_path": cb_leg,
            "other_configs": other,
        }
        print(json.dumps(line), flush=True)

    op.close()
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
