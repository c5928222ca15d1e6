#!/bin/bash
# Round-3 profile collection (GPU box, through gpurun).  Every rocprofv3 command has the program right after `--`
# (no env / bash -c hop), counters are collected in their own passes, and everything lands in gpurun_out/prof_r03/;
# the summaries that are to be judged are copied into profiles/ by hand afterwards.
# Usage: tools/r03_profiles.sh [stats] [pmc] [marker] [ops]
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_r03
mkdir -p "$OUT"
export TMPDIR=/tmp
ARGS=" $* "
has() { [[ "$ARGS" == *" $1 "* ]]; }
collect() {  # collect NAME DIR : copy the csv summaries of one rocprofv3 run
  for f in $(find "$2" -name "*kernel_stats.csv" -o -name "*marker_api_trace.csv" -o -name "*counter_collection.csv" -o -name "*domain_stats.csv" | head -20); do
    cp "$f" "$OUT/r03_$1_$(basename "$f" | sed 's/^[0-9]*_//')"
  done
}
if has stats; then
  for cfg in "c3:--workload c3" "c2:--workload c2" "c2lattice:--workload c2 --operator lattice" "c5:--workload c5" "c5lattice:--workload c5 --operator lattice"; do
    name=${cfg%%:*}; opts=${cfg#*:}
    d=$OUT/raw_stats_$name; rm -rf "$d"
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -o "$name" -- python3 bench.py $opts --steps 3 --warmup 1 --cpu-window 0 --no-spmv-variants > "$OUT/r03_bench_${name}_under_rocprof.json" 2> "$OUT/${name}_stats.err"
    echo "stats $name rc=$?"; collect "$name" "$d"
  done
fi
if has ops; then
  d=$OUT/raw_stats_ops; rm -rf "$d"
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -o ops -- python3 tools/operator_bench.py > "$OUT/r03_operator_bench.json" 2> "$OUT/ops.err"
  echo "ops rc=$?"; collect ops "$d"
fi
if has pmc; then
  for cfg in "c3:--workload c3" "c2lattice:--workload c2 --operator lattice" "c5:--workload c5"; do
    name=${cfg%%:*}; opts=${cfg#*:}
    for ctr in FETCH_SIZE WRITE_SIZE; do
      tag=$( [ $ctr = FETCH_SIZE ] && echo pmc_fetch || echo pmc_write )
      d=$OUT/raw_${tag}_$name; rm -rf "$d"
      timeout 900 rocprofv3 --pmc $ctr --output-format csv -d "$d" -o "$tag" -- python3 bench.py $opts --steps 2 --warmup 1 --cpu-window 0 --no-phase-timers --no-spmv-variants > "$OUT/${name}_${tag}.json" 2> "$OUT/${name}_${tag}.err"
      echo "pmc $name $ctr rc=$?"
      f=$(find "$d" -name "*counter_collection.csv" | head -1)
      [ -n "$f" ] && cp "$f" "$OUT/r03_${name}_${tag}_counter_collection.csv"
    done
    mkdir -p "$OUT/pmc_$name"
    cp "$OUT/r03_${name}_pmc_fetch_counter_collection.csv" "$OUT/pmc_$name/pmc_fetch_counter_collection.csv" 2>/dev/null
    cp "$OUT/r03_${name}_pmc_write_counter_collection.csv" "$OUT/pmc_$name/pmc_write_counter_collection.csv" 2>/dev/null
    nn=10000000; [ $name != c3 ] && nn=1000000
    python3 tools/pmc_summary.py "$OUT/pmc_$name" "$OUT/r03_${name}_pmc_traffic.json" $nn 6 | tail -12
  done
  d=$OUT/raw_pmc_ops; 
  for ctr in FETCH_SIZE WRITE_SIZE; do
    tag=$( [ $ctr = FETCH_SIZE ] && echo pmc_fetch || echo pmc_write )
    rm -rf "$d$tag"
    timeout 600 rocprofv3 --pmc $ctr --output-format csv -d "$d$tag" -o "$tag" -- python3 tools/operator_bench.py > /dev/null 2> "$OUT/ops_${tag}.err"
    f=$(find "$d$tag" -name "*counter_collection.csv" | head -1)
    mkdir -p "$OUT/pmc_ops"; [ -n "$f" ] && cp "$f" "$OUT/pmc_ops/${tag}_counter_collection.csv"
  done
  python3 tools/pmc_summary.py "$OUT/pmc_ops" "$OUT/r03_ops_pmc_traffic.json" 16777216 | tail -8
fi
if has marker; then
  d=$OUT/raw_marker_c3; rm -rf "$d"
  timeout 600 rocprofv3 --marker-trace --kernel-trace --stats --output-format csv -d "$d" -o marker -- python3 bench.py --steps 1 --warmup 0 --cpu-window 0 --window 30 --spmv-reps 2 > "$OUT/marker_bench.json" 2> "$OUT/marker.err"
  echo "marker rc=$?"; collect marker "$d"; ls "$d" -R | head -20
fi
rm -rf "$OUT"/raw_*   # keep the summaries only (gpurun_out is capped)
ls -la "$OUT" | head -60
