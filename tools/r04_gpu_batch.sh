#!/bin/bash
# Round-4 measurement batch (run on the GPU box through gpurun; everything lands in gpurun_out/, summaries are copied to
# profiles/ by hand).  Usage: tools/r04_gpu_batch.sh [tests] [final] [kstats] [pmc] [shard] ...
# Every command reads stdin from /dev/null and sits under `timeout`: a tool waiting for input must not eat the budget.
set -u
mkdir -p gpurun_out
ARGS=" $* "
has() { [[ "$ARGS" == *" $1 "* ]]; }
SHA=${LL_PROFILE_HEAD:-unknown}
export TMPDIR=/tmp
if has tests; then
  timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider --durations=8 > gpurun_out/r04_tests_at_$SHA.log 2>&1 < /dev/null
  echo "tests rc=$?"; tail -14 gpurun_out/r04_tests_at_$SHA.log
fi
line() {
  python3 - "$1" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("   value %.1f it/s  host_io %.1f  ms/step %.3f  spmv %.4f ms (frac %.3f)  orth frac %s  cpu %s" % (d["value"], d["value_host_io"], d["ms_per_step"], d["spmv"]["ms"], d["roofline"]["frac"], d["roofline_orth"]["frac"], (d.get("cpu_baseline") or {}).get("value")))
PY
}
if has final; then
  for cfg in "default:" "c2:--workload c2" "c2lattice:--workload c2 --operator lattice --cpu-window 0" "c5:--workload c5" "c3band:--workload c3band --cpu-window 0" "c3_steps20:--workload c3 --steps 20 --warmup 2 --cpu-window 0 --no-spmv-variants --no-other-configs"; do
    name=${cfg%%:*}; opts=${cfg#*:}
    timeout 900 python3 bench.py $opts > gpurun_out/r04_final_bench_$name.json 2> gpurun_out/r04_final_bench_$name.err < /dev/null; echo "final bench $name rc=$?"
    line gpurun_out/r04_final_bench_$name.json
  done
fi
if has kstats; then
  for cfg in ${KSTATS_CFGS:-"c3:--workload c3"}; do
    name=${cfg%%:*}; opts=${cfg#*:}
    d=/tmp/prof_r04_ks_$name; rm -rf $d
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o ks -- python3 bench.py $opts --steps 3 --warmup 1 --cpu-window 0 --no-spmv-variants --no-other-configs > gpurun_out/r04_bench_${name}_under_rocprof.json 2> gpurun_out/r04_ks_$name.err < /dev/null
    echo "kstats $name rc=$?"
    f=$(find $d -name "*kernel_stats.csv" | head -1)
    if [ -n "$f" ]; then cp "$f" gpurun_out/r04_${name}_kernel_stats.csv; python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:-float(r['TotalDurationNs']))
for r in rows[:10]: print(r['Name'][:80].ljust(82), r['Calls'].rjust(6), ('%.1f'%(float(r['AverageNs'])/1e3)).rjust(8), 'us')
PY
    fi
  done
fi
if has pmc; then
  name=c3
  d=/tmp/prof_r04_pmc_$name; rm -rf $d; mkdir -p $d
  for ctr in fetch:FETCH_SIZE write:WRITE_SIZE; do
    tag=${ctr%%:*}; c=${ctr#*:}
    timeout 900 rocprofv3 --pmc $c --output-format csv -d $d -o pmc_$tag -- python3 bench.py --workload c3 --steps 2 --warmup 1 --cpu-window 0 --no-spmv-variants --no-other-configs --no-phase-timers > gpurun_out/r04_pmc_$tag.json 2> gpurun_out/r04_pmc_$tag.err < /dev/null
    echo "pmc $tag rc=$?"
    f=$(find $d -name "pmc_${tag}_counter_collection.csv" | head -1)
    [ -n "$f" ] && cp "$f" $d/pmc_${tag}_counter_collection.csv 2>/dev/null
  done
  ls $d | head; python3 tools/pmc_summary.py $d gpurun_out/r04_c3_pmc_traffic.json 10000000 3 100 && python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r04_c3_pmc_traffic.json'))
for k,v in d['kernels'].items():
    if any(s in k for s in ('pb_phase','lagged_kernel','scale_kernel','mdot_kernel')): print(k[:60], v['launches'], round(v['fetch_bytes_mean']/1e9,3), round(v['write_bytes_mean']/1e9,3))
print({k:v for k,v in d.items() if k not in ('kernels',)})
PY
fi
if has shard; then
  timeout 600 python3 tools/shard_compute_probe.py 1 2 4 8 > gpurun_out/r04_shard_compute_probe.txt 2>&1 < /dev/null; cat gpurun_out/r04_shard_compute_probe.txt
fi
