#!/bin/bash
# Round-6 measurement batch (run on the GPU box through gpurun; everything lands in gpurun_out/, the summaries that are to be
# judged are copied to profiles/ afterwards by tools/r06_collect.sh).
# Usage: tools/r06_gpu_batch.sh [tests] [final] [conv] [profiles] [window500] [shard] [firstcall]
# Every command reads stdin from /dev/null and sits under `timeout`.
set -u
mkdir -p gpurun_out
ARGS=" $* "
has() { [[ "$ARGS" == *" $1 "* ]]; }
export TMPDIR=/tmp
if has tests; then
  timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider --durations=8 > gpurun_out/r06_tests_final.log 2>&1 < /dev/null
  echo "tests rc=$?"; tail -14 gpurun_out/r06_tests_final.log
  LL_BLAS_SMALL_BYTES=0 timeout 1200 python -m pytest tests/test_gpu_engines.py tests/test_gpu_round3.py tests/test_gpu_fuzz.py tests/test_gpu_float.py \
      tests/test_gpu_long_runs.py tests/test_gpu_pair.py -m gpu -x -q -p no:cacheprovider -k "not small_vector_geometry" > gpurun_out/r06_tests_final_streaming.log 2>&1 < /dev/null
  echo "streaming-geometry tests rc=$?"; tail -4 gpurun_out/r06_tests_final_streaming.log
  LL_PAIR_GS=0 timeout 900 python -m pytest tests/test_gpu_round2.py tests/test_gpu_round3.py -m gpu -x -q -p no:cacheprovider \
      -k "(c3_full or c2_full or lagged) and not real_reference_fixture" > gpurun_out/r06_tests_final_pair_off.log 2>&1 < /dev/null
  echo "LL_PAIR_GS=0 tests rc=$?"; tail -3 gpurun_out/r06_tests_final_pair_off.log
fi
line() {
  python3 - "$1" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d["roofline"]
print("   value %.1f it/s  host_io %.1f  ms/step %.3f  spmv %.4f ms (frac %.3f, of measured read %.3f)  orth frac %s  cpu %s" % (d["value"], d["value_host_io"], d["ms_per_step"], d["spmv"]["ms"], r["frac"], r.get("frac_of_measured", 0), d["roofline_orth"]["frac"], (d.get("cpu_baseline") or {}).get("value")))
PY
}
if has final; then
  for cfg in "default:" "c2:--workload c2" "c2lattice:--workload c2 --operator lattice --cpu-window 0" "c5:--workload c5" "c3band:--workload c3band --cpu-window 0" \
             "c3_steps20:--workload c3 --steps 20 --warmup 2 --cpu-window 0 --no-spmv-variants --no-other-configs" \
             "n1e5:--workload c2 --size 316 --steps 20 --warmup 2 --cpu-window 0" "n1e4:--workload c2 --size 100 --steps 20 --warmup 2 --cpu-window 0" \
             "c3_pair_off:--workload c3 --cpu-window 0 --no-spmv-variants --no-other-configs"; do
    name=${cfg%%:*}; opts=${cfg#*:}
    if [ "$name" = c3_pair_off ]; then export LL_PAIR_GS=0; else unset LL_PAIR_GS; fi
    timeout 900 python3 bench.py $opts > gpurun_out/r06_final_bench_$name.json 2> gpurun_out/r06_final_bench_$name.err < /dev/null; echo "final bench $name rc=$?"
    line gpurun_out/r06_final_bench_$name.json
  done
  unset LL_PAIR_GS
fi
if has window500; then
  timeout 900 python3 bench.py --window 500 --steps 3 --warmup 1 --cpu-window 0 --no-spmv-variants --no-other-configs > gpurun_out/r06_final_bench_c3_window500.json 2> gpurun_out/r06_w500.err < /dev/null
  echo "window 500 rc=$?"; line gpurun_out/r06_final_bench_c3_window500.json
  LL_PAIR_GS=0 timeout 900 python3 bench.py --window 500 --steps 3 --warmup 1 --cpu-window 0 --no-spmv-variants --no-other-configs > gpurun_out/r06_final_bench_c3_window500_pair_off.json 2> gpurun_out/r06_w500b.err < /dev/null
  echo "window 500 (LL_PAIR_GS=0) rc=$?"; line gpurun_out/r06_final_bench_c3_window500_pair_off.json
fi
if has conv; then
  for wl in c3 c2; do
    timeout 600 python3 tests/convergence_run.py $wl > gpurun_out/r06_convergence_${wl}_defaults.json 2> gpurun_out/r06_convergence_$wl.err < /dev/null
    echo "convergence $wl rc=$?"; tail -c 600 gpurun_out/r06_convergence_${wl}_defaults.json
  done
fi
if has firstcall; then
  # the first run() of a process that FOLLOWS other processes on the same GPU (freshly released VRAM: ~120 ms per 4 GiB hipMalloc)
  for i in 1 2 3 4 5 6 7; do
    timeout 300 python3 bench.py --steps 2 --warmup 1 --cpu-window 0 --no-spmv-variants --no-other-configs --no-phase-timers > /dev/null 2>&1 < /dev/null
  done
  timeout 600 python3 tests/convergence_run.py c3 > gpurun_out/r06_convergence_c3_after_seven_processes.json 2> gpurun_out/r06_firstcall.err < /dev/null
  echo "first call after seven processes rc=$?"; tail -c 700 gpurun_out/r06_convergence_c3_after_seven_processes.json
fi
if has shard; then
  { echo "# tools/shard_compute_probe.py (one GPU plays rank 0 of N; the exchange is a local stand-in: COMPUTE only), default chunks and LL_GATHER_CHUNKS=1"
    timeout 900 python3 tools/shard_compute_probe.py 1 2 4 8 < /dev/null
    echo "# LL_GATHER_CHUNKS=1 (own blocks, then ALL remote blocks in one phase-1 launch)"
    LL_GATHER_CHUNKS=1 timeout 900 python3 tools/shard_compute_probe.py 2 4 8 < /dev/null
  } > gpurun_out/r06_shard_compute_probe.txt 2> gpurun_out/r06_shard.err
  echo "shard probe rc=$?"; cut -c1-260 gpurun_out/r06_shard_compute_probe.txt
fi
if has profiles; then
  bash tools/r06_profiles.sh stats pmc > gpurun_out/r06_profiles.log 2>&1 < /dev/null
  echo "profiles rc=$?"; grep -n "rc=\|calibration\|pb_phase\|tl_spmv\|pair_sweep\|pair_three\|orth_bytes" gpurun_out/r06_profiles.log | cut -c1-330
fi
