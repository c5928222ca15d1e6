#!/bin/bash
# Round-5 measurement batch (run on the GPU box through gpurun; everything lands in gpurun_out/, the summaries that are to be
# judged are copied to profiles/ afterwards).  Usage: tools/r05_gpu_batch.sh [tests] [final] [conv] [profiles]
# Every command reads stdin from /dev/null and sits under `timeout`.
set -u
mkdir -p gpurun_out
ARGS=" $* "
has() { [[ "$ARGS" == *" $1 "* ]]; }
export TMPDIR=/tmp
if has tests; then
  timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider --durations=8 > gpurun_out/r05_tests_final.log 2>&1 < /dev/null
  echo "tests rc=$?"; tail -14 gpurun_out/r05_tests_final.log
  LL_BLAS_SMALL_BYTES=0 timeout 1200 python -m pytest tests/test_gpu_engines.py tests/test_gpu_round3.py tests/test_gpu_fuzz.py tests/test_gpu_float.py \
      tests/test_gpu_long_runs.py tests/test_gpu_pair.py -m gpu -x -q -p no:cacheprovider > gpurun_out/r05_tests_final_streaming.log 2>&1 < /dev/null
  echo "streaming-geometry tests rc=$?"; tail -4 gpurun_out/r05_tests_final_streaming.log
  LL_PAIR_GS=0 timeout 900 python -m pytest tests/test_gpu_round2.py tests/test_gpu_round3.py -m gpu -x -q -p no:cacheprovider \
      -k "c3_full or c2_full or lagged" > gpurun_out/r05_tests_final_pair_off.log 2>&1 < /dev/null
  echo "LL_PAIR_GS=0 tests rc=$?"; tail -3 gpurun_out/r05_tests_final_pair_off.log
fi
line() {
  python3 - "$1" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("   value %.1f it/s  host_io %.1f  ms/step %.3f  spmv %.4f ms (frac %.3f)  orth frac %s  cpu %s" % (d["value"], d["value_host_io"], d["ms_per_step"], d["spmv"]["ms"], d["roofline"]["frac"], d["roofline_orth"]["frac"], (d.get("cpu_baseline") or {}).get("value")))
PY
}
if has final; then
  for cfg in "default:" "c2:--workload c2" "c2lattice:--workload c2 --operator lattice --cpu-window 0" "c5:--workload c5" "c3band:--workload c3band --cpu-window 0" \
             "c3_steps20:--workload c3 --steps 20 --warmup 2 --cpu-window 0 --no-spmv-variants --no-other-configs" \
             "c3_pair_off:--workload c3 --cpu-window 0 --no-spmv-variants --no-other-configs"; do
    name=${cfg%%:*}; opts=${cfg#*:}
    if [ "$name" = c3_pair_off ]; then export LL_PAIR_GS=0; else unset LL_PAIR_GS; fi
    timeout 900 python3 bench.py $opts > gpurun_out/r05_final_bench_$name.json 2> gpurun_out/r05_final_bench_$name.err < /dev/null; echo "final bench $name rc=$?"
    line gpurun_out/r05_final_bench_$name.json
  done
  unset LL_PAIR_GS
fi
if has conv; then
  for wl in c3 c2; do
    timeout 600 python3 tests/convergence_run.py $wl > gpurun_out/r05_convergence_${wl}_defaults.json 2> gpurun_out/r05_convergence_$wl.err < /dev/null
    echo "convergence $wl rc=$?"; tail -c 600 gpurun_out/r05_convergence_${wl}_defaults.json
  done
fi
if has profiles; then
  bash tools/r05_profiles.sh stats pmc > gpurun_out/r05_profiles.log 2>&1 < /dev/null
  echo "profiles rc=$?"; grep -n "rc=\|calibration\|pb_phase\|tl_spmv\|pair_sweep\|pair_three\|orth_bytes" gpurun_out/r05_profiles.log | cut -c1-330
fi
