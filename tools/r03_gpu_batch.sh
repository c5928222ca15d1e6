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
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r03/sweep -o sweep -- python3 tools/spmv_sweep.py --variants "$V" --rounds 5 --repeat-check 2 > gpurun_out/r3_sweep.jsonl 2> gpurun_out/r3_sweep.err; echo "sweep rc=$?"
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
