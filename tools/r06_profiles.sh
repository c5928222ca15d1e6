#!/bin/bash
# Round-6 profile collection (GPU box, through gpurun).  Every rocprofv3 command has the program right after `--`
# (no env / bash -c hop), counters are collected in their own passes (--pmc alone), and everything lands in
# gpurun_out/prof_r06/; the summaries that are to be judged are copied into profiles/ afterwards.
# Usage: tools/r06_profiles.sh [stats] [pmc]   (workloads: c3 = the metric's configuration, c3band = its banded variant)
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_r06
mkdir -p "$OUT"
export TMPDIR=/tmp
ARGS=" $* "
has() { [[ "$ARGS" == *" $1 "* ]]; }
collect() {
  for f in $(find "$2" -name "*kernel_stats.csv" | head -5); do
    cp "$f" "$OUT/r06_$1_$(basename "$f" | sed 's/^[0-9]*_//')"
  done
}
if has stats; then
  for cfg in "c3:--workload c3" "c3band:--workload c3band" "c2:--workload c2"; do
    name=${cfg%%:*}; opts=${cfg#*:}
    d=$OUT/raw_stats_$name; rm -rf "$d"
    timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -o "$name" -- python3 bench.py $opts --steps 3 --warmup 1 --cpu-window 0 --no-spmv-variants --no-other-configs > "$OUT/r06_bench_${name}_under_rocprof.json" 2> "$OUT/${name}_stats.err"
    echo "stats $name rc=$?"; collect "$name" "$d"
  done
fi
if has pmc; then
  for cfg in "c3:--workload c3" "c3band:--workload c3band"; do
    name=${cfg%%:*}; opts=${cfg#*:}
    for ctr in FETCH_SIZE WRITE_SIZE; do
      tag=$( [ $ctr = FETCH_SIZE ] && echo pmc_fetch || echo pmc_write )
      d=$OUT/raw_${tag}_$name; rm -rf "$d"
      timeout 900 rocprofv3 --pmc $ctr --output-format csv -d "$d" -o "$tag" -- python3 bench.py $opts --steps 2 --warmup 1 --cpu-window 0 --no-phase-timers --no-spmv-variants --no-other-configs > "$OUT/${name}_${tag}.json" 2> "$OUT/${name}_${tag}.err"
      echo "pmc $name $ctr rc=$?"
      f=$(find "$d" -name "*counter_collection.csv" | head -1)
      mkdir -p "$OUT/pmc_$name"
      [ -n "$f" ] && cp "$f" "$OUT/pmc_$name/${tag}_counter_collection.csv"
    done
    # windows of 100 iterations the profiled command runs: warmup 1 + steps 2, and the same three through the other I/O boundary
    python3 tools/pmc_summary.py "$OUT/pmc_$name" "$OUT/r06_${name}_pmc_traffic.json" 10000000 6 | tail -14
  done
fi
rm -rf "$OUT"/raw_* "$OUT"/pmc_*/pmc_*_counter_collection.csv   # keep the summaries only (gpurun_out is capped)
ls -la "$OUT" | head -40
