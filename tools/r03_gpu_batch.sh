#!/bin/bash
# Round-3 measurement batch (run on the GPU box through gpurun; everything lands in gpurun_out/).
# Usage: tools/r03_gpu_batch.sh [tests] [tests_pb] [sweep] [bench] ...
set -u
mkdir -p gpurun_out
ARGS=" $* "
has() { [[ "$ARGS" == *" $1 "* ]]; }
if has tests; then
  timeout 1500 python -m pytest tests -m gpu -q --maxfail=12 -p no:cacheprovider > gpurun_out/r3_tests.log 2>&1; echo "tests rc=$?"; tail -25 gpurun_out/r3_tests.log
fi
if has tests_pb; then
  timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_round3.py tests/test_gpu_round2.py tests/test_gpu_multirank.py tests/test_gpu_float.py -m gpu -q --maxfail=12 -p no:cacheprovider > gpurun_out/r3_tests_pb.log 2>&1; echo "tests_pb rc=$?"; tail -25 gpurun_out/r3_tests_pb.log
fi
if has sweep; then
  V="${SWEEP_VARIANTS:-pre_d33:LL_SPMV_KERNEL=pb}"
  export TMPDIR=/tmp
  rm -rf gpurun_out/prof_r03/sweep
  timeout 900 ${SWEEP_PREFIX:-rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r03/sweep -o sweep --} python3 tools/spmv_sweep.py --variants "$V" --rounds 5 --repeat-check 2 > gpurun_out/r3_sweep.jsonl 2> gpurun_out/r3_sweep.err; echo "sweep rc=$?"
  python - <<'PY'
import json
for l in open('gpurun_out/r3_sweep.jsonl'):
    d=json.loads(l); print(d['variant'], round(d['ms_median'],4), round(d['ms_min'],4), d.get('bit_identical_over_4_launches'), d['max_abs_dev_from_first_variant'])
PY
  tail -3 gpurun_out/r3_sweep.err
  python3 - <<'PY'
import csv,glob
for f in glob.glob('gpurun_out/prof_r03/sweep/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'pb_phase' in r['Name'] or 'absmax' in r['Name'] or 'spmv' in r['Name']: print(r['Name'][:64].ljust(66), r['Calls'], r['AverageNs'], r['MinNs'])
PY
fi
if has bench; then
  timeout 900 python bench.py > gpurun_out/r3_bench_default.json 2> gpurun_out/r3_bench_default.err; echo "bench rc=$?"; cut -c1-1500 gpurun_out/r3_bench_default.json; tail -3 gpurun_out/r3_bench_default.err
fi
if has bench_all; then
  for cfg in ${BENCH_CFGS:-"c3:--workload c3"} "c2:--workload c2" "c2lattice:--workload c2 --operator lattice" "c5:--workload c5" "c5lattice:--workload c5 --operator lattice" "n1e4:--workload c2 --size 100" "n1e5:--workload c2 --size 316"; do
    name=${cfg%%:*}; opts=${cfg#*:}
    timeout 600 python bench.py $opts --cpu-window 0 > gpurun_out/r3_bench_$name.json 2> gpurun_out/r3_bench_$name.err; echo "bench $name rc=$?"
    python - gpurun_out/r3_bench_$name.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("   value %.1f it/s  host_io %.1f  ms/step %.3f  spmv %.4f ms (frac %.3f)  orth frac %s" % (d["value"], d["value_host_io"], d["ms_per_step"], d["spmv"]["ms"], d["roofline"]["frac"], d["roofline_orth"]["frac"]))
PY
  done
fi
if has kstats; then
  export TMPDIR=/tmp
  for cfg in ${KSTATS_CFGS:-"c3:--workload c3" "c5:--workload c5" "c2:--workload c2"}; do
    name=${cfg%%:*}; opts=${cfg#*:}
    mkdir -p gpurun_out/prof_r03; d=gpurun_out/prof_r03/ks_$name; rm -rf $d
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o ks -- python3 bench.py $opts --steps 3 --warmup 1 --cpu-window 0 --no-spmv-variants > gpurun_out/prof_r03/ks_$name.json 2> gpurun_out/prof_r03/ks_$name.err
    echo "kstats $name rc=$?"
    python3 - $d <<'PY'
import csv,glob,sys
for f in glob.glob(sys.argv[1]+'/**/*kernel_stats.csv', recursive=True):
    rows=list(csv.DictReader(open(f)))
    rows.sort(key=lambda r:-float(r['TotalDurationNs']))
    for r in rows[:10]: print(r['Name'][:80].ljust(82), r['Calls'].rjust(6), ('%.1f'%(float(r['AverageNs'])/1e3)).rjust(8), 'us')
PY
  done
fi
if has kgaps; then
  export TMPDIR=/tmp
  mkdir -p gpurun_out/prof_r03
  for cfg in "n1e4:--workload c2 --size 100" "n1e5:--workload c2 --size 316"; do
    name=${cfg%%:*}; opts=${cfg#*:}
    for fuse in 1 0; do
      d=gpurun_out/prof_r03/kg_${name}_f$fuse; rm -rf $d
      LL_FUSE_LAUNCHES=$fuse timeout 600 rocprofv3 --kernel-trace --output-format csv -d $d -o kg -- python3 bench.py $opts --steps 5 --warmup 1 --cpu-window 0 --no-spmv-variants --no-phase-timers > $d.json 2> $d.err
      echo "kgaps $name fuse=$fuse rc=$? value $(python3 -c "import json;print(json.loads(open('$d.json').read().strip().splitlines()[-1])['value'])")"
      python3 tools/trace_gaps2.py $(find $d -name "*kernel_trace.csv" | head -1)
    done
  done
fi
if has final; then
  for cfg in "default:" "c3:--workload c3 --cpu-window 0" "c2:--workload c2" "c2lattice:--workload c2 --operator lattice --cpu-window 0" "c5:--workload c5" "c5lattice:--workload c5 --operator lattice --cpu-window 0" "c3band:--workload c3band --cpu-window 0" "c3_window500:--workload c3 --window 500 --steps 2 --cpu-window 0 --no-spmv-variants" "n1e4:--workload c2 --size 100 --cpu-window 0" "n1e5:--workload c2 --size 316 --cpu-window 0" "c3_steps20:--workload c3 --steps 20 --warmup 2 --cpu-window 0 --no-spmv-variants"; do
    name=${cfg%%:*}; opts=${cfg#*:}
    timeout 900 python bench.py $opts > gpurun_out/r03_final_bench_$name.json 2> gpurun_out/r03_final_bench_$name.err; echo "final bench $name rc=$?"
    python - gpurun_out/r03_final_bench_$name.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("   value %.1f it/s  host_io %.1f  ms/step %.3f  spmv %.4f ms (frac %.3f)  orth frac %s  cpu %s" % (d["value"], d["value_host_io"], d["ms_per_step"], d["spmv"]["ms"], d["roofline"]["frac"], d["roofline_orth"]["frac"], (d.get("cpu_baseline") or {}).get("value")))
PY
  done
  for wl in c3 c2; do
    timeout 900 python tests/convergence_run.py $wl > gpurun_out/r03_convergence_${wl}_defaults.json 2> gpurun_out/r03_convergence_${wl}.err; echo "convergence $wl rc=$?"; cat gpurun_out/r03_convergence_${wl}_defaults.json | cut -c1-600
  done
fi
