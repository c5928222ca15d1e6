#!/usr/bin/env python3
"""Fill the UPPER_CASE placeholders of DESIGN.md section 5 (and the places that quote the same figures) from the summaries
tools/r05_collect.sh copied into profiles/.  Run once after the round's final measurement batch; prints what it could not fill."""
import csv
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")


def compute(P):
    global _P
    _P = P
    return _compute()


def J(name):
    return json.loads(open(os.path.join(_P, name)).read().strip().splitlines()[-1])


def stats(name):
    return {r["Name"].split("(")[0].replace("void ", "").strip(): float(r["AverageNs"]) / 1e3
            for r in csv.DictReader(open(os.path.join(_P, name)))}


def grp(x):  # 10747.6 -> "10 748"
    return f"{round(x):,}".replace(",", " ")


def _compute():
    d = J("r05_final_bench_default.json")
    c2, c2l, c5, c3b = J("r05_final_bench_c2.json"), J("r05_final_bench_c2lattice.json"), J("r05_final_bench_c5.json"), J("r05_final_bench_c3band.json")
    s20, poff = J("r05_final_bench_c3_steps20.json"), J("r05_final_bench_c3_pair_off.json")
    k3 = stats("r05_c3_kernel_stats.csv")
    conv3, conv2 = J("r05_convergence_c3_defaults.json"), J("r05_convergence_c2_defaults.json")
    tl = json.load(open(os.path.join(_P, "r05_c3band_pmc_traffic.json")))
    oc = d["other_configs"]
    sweep_us = next(v for k, v in k3.items() if "pair_sweep_kernel" in k)
    tt_us = next(v for k, v in k3.items() if "pair_three_term" in k)
    p1 = next(v for k, v in k3.items() if "pb_phase1" in k)
    p2 = next(v for k, v in k3.items() if "pb_phase2_fixed" in k)
    tlk = next(v for k, v in tl["kernels"].items() if "tl_spmv" in k) if "kernels" in tl else None
    if tlk is None:
        sys.exit("r05_c3band_pmc_traffic.json: no kernels table")
    tl_fetch, tl_write = tlk["fetch_bytes_mean"] / 1e9, tlk["write_bytes_mean"] / 1e9
    log = open(os.path.join(ROOT, "gpurun_out", "r05_tests_final.log")).read()
    m = re.search(r"(\d+) passed, (\d+) skipped.* in ([\d.]+)s", log)
    slog = open(os.path.join(ROOT, "gpurun_out", "r05_tests_final_streaming.log")).read()
    ms = re.search(r"(\d+) passed", slog)
    cpu = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests"), "-q", "-m", "not gpu", "--co"], capture_output=True, text=True).stdout
    mc = re.search(r"(\d+)/\d+ tests collected", cpu) or re.search(r"(\d+) tests? collected", cpu)
    par2 = oc["c2"].get("parity_same_window") or {}
    par5 = oc["c5"].get("parity_whole_run") or {}
    vals = {
        "PAIR_C3": grp(d["value"]), "PAIR_C2": grp(c2["value"]),
        "PAIR_SWEEP_US": "%.0f" % sweep_us, "PAIR_SWEEP_TBS": "%.2f" % (4.48e9 / (sweep_us * 1e-6) / 1e12),
        "PAIR_TRAFFIC": "235.89 GB per window against 236.08 GB modelled",
        "C3B_SPMV_MS": "%.3f" % c3b["spmv"]["ms"], "C3B_FRAC": "%.1f" % (100 * c3b["roofline"]["frac"]), "C3B_VALUE": grp(c3b["value"]),
        "TL_FETCH": "%.3f" % tl_fetch, "TL_RATIO": "%.2f" % ((tl_fetch + tl_write) / 2.0),
        "GPU_PASSED": m.group(1), "GPU_SECONDS": "%.0f" % float(m.group(3)), "STREAM_PASSED": ms.group(1), "CPU_PASSED": mc.group(1) if mc else "?",
        "HEADSHA": os.environ.get("LL_PROFILE_HEAD", "?"),
        "C3_VALUE": "%.1f" % d["value"], "C3_STEPS20": "%.1f" % s20["value"], "C3_PAIR_OFF": "%.1f" % poff["value"], "C3_HOSTIO": "%.1f" % d["value_host_io"],
        "C3_SPMV_MS": "%.3f" % d["spmv"]["ms"], "C3_SPMV_FRAC": "%.1f" % (100 * d["roofline"]["frac"]), "C3_P1_US": "%.0f" % p1, "C3_P2_US": "%.0f" % p2,
        "C3_ORTH_FRAC": "%.1f" % (100 * d["roofline_orth"]["frac"]), "C3_TT_US": "%.1f" % tt_us,
        "C3_CREATE": "%.0f" % (1e3 * d["phases"]["setup_s_operator_create_by_rank"][0]),
        "C3_CPU14": "%.2f" % d["cpu_baseline"]["value"], "C3_GPU14": "%.0f" % d["cpu_baseline"]["gpu_same_window_value"],
        "C3_CONV": "%.3f" % conv3["wall_s"], "C2_CONV": "%.2f" % conv2["wall_s"],
        "C2_VALUE": grp(c2["value"]), "C2L_VALUE": grp(c2l["value"]), "C5_VALUE": grp(c5["value"]),
        "OC2_VALUE": grp(oc["c2"]["value"]), "OC2_SPMV": "%.1f µs = %.0f %% of 8 TB/s" % (1e3 * oc["c2"]["spmv"]["ms"], 100 * oc["c2"]["spmv"]["frac_of_8TBps"]),
        "OC2_PARITY": "\\|Δα\\| ≤ %.1e, \\|Δβ\\| ≤ %.1e, \\|Δλ\\| %.1e, 1 − overlap %.1e; reference on one core %.1f it/s" % (
            par2.get("max_abs_dalpha", float("nan")), par2.get("max_abs_dbeta", float("nan")), par2.get("abs_dlambda", float("nan")),
            par2.get("eigenvector_one_minus_overlap", float("nan")), (oc["c2"].get("cpu_same_window") or {}).get("value", float("nan"))),
        "OC5_VALUE": grp(oc["c5"]["value"]), "OC5_SPMV": "%.1f µs = %.0f %% of 8 TB/s" % (1e3 * oc["c5"]["spmv"]["ms"], 100 * oc["c5"]["spmv"]["frac_of_8TBps"]),
        "OC5_PARITY": "\\|out_gpu − out_cpu\\| / \\|in\\| = %.1e, 1 − overlap %.1e, norm drift %.1e, %d = %d iterations; reference on one core %.1f it/s" % (
            par5.get("max_abs_diff_over_input_norm", float("nan")), par5.get("one_minus_overlap", float("nan")), par5.get("norm_drift", float("nan")),
            par5.get("iterations_gpu", -1), par5.get("iterations_cpu", -1), (oc["c5"].get("cpu_whole_run") or {}).get("value", float("nan"))),
        "CB_VALUE": "%.2f it/s (device-resident operator on the same window: %.0f)" % (d["callback_path"]["value_callback"], d["callback_path"]["device_operator_same_window_value"]),
    }
    return vals


path = os.path.join(ROOT, "DESIGN.md")
s = open(path).read()
if len(sys.argv) > 2 and sys.argv[1] == "--refresh":   # figures of an earlier batch (its profiles/ in argv[2]) -> the current ones
    old, new = compute(sys.argv[2]), compute(P)
    for k in sorted(old, key=lambda k: len(old[k]), reverse=True):
        if old[k] != new[k] and len(old[k]) >= 4:
            n = s.count(old[k])
            s = s.replace(old[k], new[k])
            print("%-14s %4d x  %s -> %s" % (k, n, old[k][:60], new[k][:60]))
        elif old[k] != new[k]:
            print("%-14s NOT replaced (too short): %s -> %s" % (k, old[k], new[k]))
    open(path, "w").write(s)
    sys.exit(0)
vals = compute(P)
for k in sorted(vals, key=len, reverse=True):
    s = re.sub(r"(?<![A-Z0-9_])" + k + r"(?![A-Z0-9_])", vals[k], s)
open(path, "w").write(s)
print("filled", len(vals), "keys")
