#!/bin/bash
# Same-box A/B of two builds of the library (lambda-lanczos_amd/lib_base vs lib) on a bench workload: interleaved processes,
# SpMV by HIP events and the Lanczos window.  Usage: tools/ab_bench_two_builds.sh [bench.py options]
for i in 1 2 3; do
  for which in base new; do
    if [ $which = base ]; then export LL_LIB_PATH=$(pwd)/lambda-lanczos_amd/lib_base/liblanczos_hip.so; else unset LL_LIB_PATH; fi
    python3 bench.py "$@" --no-other-configs --cpu-window 0 --no-spmv-variants --steps 2 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$which', 'spmv_ms %.4f  it/s %.1f  op_s %.4f orth_s %.4f' % (d['spmv']['ms'], d['value'], d['phases']['device_s_operator'], d['phases']['device_s_orth']))"
  done
done
