#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes into per-kernel HBM traffic per launch (profiles/*_pmc_traffic.json).

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d DIR -o pmc_fetch -- python3 bench.py ...
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d DIR -o pmc_write -- python3 bench.py ...   (separate pass)
    python tools/pmc_summary.py DIR OUT.json [n] [windows] [window_iterations]

`windows` = number of LambdaLanczos::run windows the profiled command executed (bench.py: steps + warmup, with
--cpu-window 0): the per-window traffic of the Gram-Schmidt kernels (sum over their launches / windows) is what
bench.py reports as roofline_orth.traffic.

Units and corrections (MI355X_MICROARCH.md, "HBM" section): the counters are in KiB; on gfx950 FETCH_SIZE reports
exactly half of the bytes of wide coalesced streaming reads (128-B requests tallied at 64 B), so fetch bytes =
FETCH_SIZE * 1024 * 2; WRITE_SIZE is exact.  The factor is calibrated in the same pass on scale_kernel, whose traffic
is known (reads n*8 bytes, writes n*8 bytes): the JSON records the calibration ratio next to the numbers."""
import collections
import csv
import json
import sys


def load(path):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("ll::", "")
        agg[k].append(float(r["Counter_Value"]))
    return agg


def main():
    d, out = sys.argv[1], sys.argv[2]
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 10_000_000
    windows = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    window_iterations = int(sys.argv[5]) if len(sys.argv) > 5 else 100
    fetch, write = load(d + "/pmc_fetch_counter_collection.csv"), load(d + "/pmc_write_counter_collection.csv")
    import hashlib
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    for name in ("kernels.hip", "spmv_pb.hip", "dev_helpers.hpp", "fixed_round.hpp"):
        with open(os.path.join(root, "lambda-lanczos_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    res = {"kernel_sources_sha16": h.hexdigest()[:16],  # bench.py compares it with the sources it runs (roofline.traffic_age)
           "collected_at_head": os.environ.get("LL_PROFILE_HEAD"),  # git HEAD of the tree the profile was taken on (set by the caller)
           "units": "bytes per launch; fetch = FETCH_SIZE KiB * 1024 * 2 (gfx950 half-count correction), "
                    "write = WRITE_SIZE KiB * 1024",
           "calibration": {}, "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        f = fetch.get(k, [0.0])
        w = write.get(k, [0.0])
        # the predicated second-pass launches of mdot/maxpy are no-ops: report the maximum (k = window) and the mean
        res["kernels"][k] = {"launches": len(f), "fetch_bytes_mean": sum(f) / len(f) * 2048, "fetch_bytes_max": max(f) * 2048,
                             "write_bytes_mean": sum(w) / len(w) * 1024, "write_bytes_max": max(w) * 1024,
                             "fetch_bytes_sum": sum(f) * 2048, "write_bytes_sum": sum(w) * 1024}
    if windows:
        res["windows"] = windows
        res["window_iterations"] = window_iterations
        loop = ("mdot_kernel", "mdot_small_kernel", "lagged_kernel", "lagged_small_kernel", "lagged_fold_kernel", "reduce_cols_kernel",
                "pair_sweep_kernel", "pair_sweep_pipe_kernel", "pair_small_kernel", "pair_three_term_kernel", "pair_predict_kernel",
                "pair_fold_kernel")
        two_sweep = ("maxpy_kernel", "maxpy_small_kernel", "scale_kernel", "scale_publish_kernel")
        # pair form: the multi-axpy / scale kernels only complete the pending vector at the END of a pass (LoopState::pair_flush,
        # outside the per-iteration phase timers): reported separately, not part of the Gram-Schmidt sweeps' traffic
        pair_form = any(k.split("<")[0] in ("pair_sweep_kernel", "pair_sweep_pipe_kernel", "pair_small_kernel") for k in res["kernels"])
        names = loop if pair_form else loop + two_sweep
        orth = [v for k, v in res["kernels"].items() if k.split("<")[0] in names]
        if pair_form:
            fl = [v for k, v in res["kernels"].items() if k.split("<")[0] in two_sweep]
            res["end_of_pass_flush_bytes_per_window"] = sum(v["fetch_bytes_sum"] + v["write_bytes_sum"] for v in fl) / windows
        res["orth_bytes_per_window"] = sum(v["fetch_bytes_sum"] + v["write_bytes_sum"] for v in orth) / windows
    sk = res["kernels"].get("scale_kernel<double>")
    if sk:
        res["calibration"] = {"kernel": "scale_kernel<double> (reads and writes n*8 bytes, n=%d)" % n,
                              "fetch_corrected_over_known": sk["fetch_bytes_mean"] / (8.0 * n),
                              "write_over_known": sk["write_bytes_mean"] / (8.0 * n)}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res["calibration"]))
    for k, v in res["kernels"].items():
        if k.startswith(("pb_phase1<", "pb_phase2<", "pb_phase2_fixed<", "spmv_stream<", "mdot_kernel<", "maxpy_kernel<", "lagged_kernel<", "stencil", "dense_mv", "gemv_basis",
                         "tl_spmv_kernel<", "tl_xmax_kernel<", "pair_sweep_kernel<", "pair_sweep_pipe_kernel<", "pair_three_term_kernel<", "bw_read_kernel<", "bw_copy_kernel<")):
            print(k, {kk: (round(vv / 1e9, 4) if isinstance(vv, float) else vv) for kk, vv in v.items()})
    if windows:
        print("orth_bytes_per_window_GB", res["orth_bytes_per_window"] / 1e9)


if __name__ == "__main__":
    main()
