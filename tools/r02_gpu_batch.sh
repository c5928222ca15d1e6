#!/bin/bash
# Round-2 measurement batch (run on the GPU box through gpurun; everything lands in gpurun_out/).
# Usage: tools/r02_gpu_batch.sh [tests] [sweep] [small] [conv] [bench]
set -u
mkdir -p gpurun_out
want() { [[ " $* " == *" $1 "* ]]; }
ARGS=" $* "
has() { [[ "$ARGS" == *" $1 "* ]]; }
if has tests; then
  timeout 1200 python -m pytest tests -m gpu -q --maxfail=12 > gpurun_out/r2_tests.log 2>&1; echo "tests rc=$?"; tail -25 gpurun_out/r2_tests.log
fi
if has sweep; then
  V="pb_default:LL_SPMV_KERNEL=pb"
  V="$V;pb_depth2:LL_SPMV_KERNEL=pb,LL_PB_DEPTH=2"
  V="$V;pb_xprop:LL_SPMV_KERNEL=pb,LL_PB_XPROP=1"
  V="$V;pb_xprop_u21:LL_SPMV_KERNEL=pb,LL_PB_XPROP=1,LL_PB_U2=1"
  V="$V;pb_xprop_d2:LL_SPMV_KERNEL=pb,LL_PB_XPROP=1,LL_PB_DEPTH=2"
  V="$V;pb_xprop_u12:LL_SPMV_KERNEL=pb,LL_PB_XPROP=1,LL_PB_U1=2"
  V="$V;pb_xprop_atomic:LL_SPMV_KERNEL=pb,LL_PB_XPROP=1,LL_PB_PHASE2=atomic"
  V="$V;pb_atomic_u21:LL_SPMV_KERNEL=pb,LL_PB_U2=1,LL_PB_PHASE2=atomic"
  V="$V;pb_rowgroups2:LL_SPMV_KERNEL=pb,LL_PB_ROW_GROUPS=2"
  V="$V;l2g_slice18:LL_SPMV_KERNEL=l2g"
  V="$V;csr_stream:LL_SPMV_KERNEL=csr"
  timeout 900 python tools/spmv_sweep.py --variants "$V" --rounds 7 > gpurun_out/r2_sweep.jsonl 2> gpurun_out/r2_sweep.err; echo "sweep rc=$?"
  cut -c1-330 gpurun_out/r2_sweep.jsonl; tail -3 gpurun_out/r2_sweep.err
fi
if has calib; then
  # the same two variants at several positions of the creation order: separates the kernel effect from the placement
  # of the operator's buffers (the first operator created in a process has shown 3-5 % slower SpMVs)
  V="d3_a:LL_SPMV_KERNEL=pb,LL_PB_DEPTH=3;d2_a:LL_SPMV_KERNEL=pb,LL_PB_DEPTH=2;d3_b:LL_SPMV_KERNEL=pb,LL_PB_DEPTH=3;d2_b:LL_SPMV_KERNEL=pb,LL_PB_DEPTH=2"
  V="$V;atomic_a:LL_SPMV_KERNEL=pb,LL_PB_PHASE2=atomic,LL_PB_U2=1;d3_c:LL_SPMV_KERNEL=pb,LL_PB_DEPTH=3;d2_c:LL_SPMV_KERNEL=pb,LL_PB_DEPTH=2;atomic_b:LL_SPMV_KERNEL=pb,LL_PB_PHASE2=atomic,LL_PB_U2=1"
  V="$V;token_a:LL_SPMV_KERNEL=pb,LL_PB_PHASE2=token;token_d2:LL_SPMV_KERNEL=pb,LL_PB_PHASE2=token,LL_PB_DEPTH=2;token_u1:LL_SPMV_KERNEL=pb,LL_PB_PHASE2=token,LL_PB_U2=1;d3_d:LL_SPMV_KERNEL=pb,LL_PB_DEPTH=3;token_b:LL_SPMV_KERNEL=pb,LL_PB_PHASE2=token;atomic_c:LL_SPMV_KERNEL=pb,LL_PB_PHASE2=atomic,LL_PB_U2=1"
  timeout 900 python tools/spmv_sweep.py --variants "$V" --rounds 7 > gpurun_out/r2_calib.jsonl 2> gpurun_out/r2_calib.err; echo "calib rc=$?"
  python -c "
import json
for l in open('gpurun_out/r2_calib.jsonl'):
    d=json.loads(l); print(d['variant'], round(d['ms_median'],4), round(d['ms_min'],4), d['bit_identical_over_4_launches'])"
fi
if has small; then
  # host tridiagonal step off the enqueueing thread: window-100 iterations/s, before (inline QR = round 1 default;
  # inline AUTO) and after (AUTO on the helper thread = round 2 default)
  : > gpurun_out/r2_small.jsonl
  for size in 100 316 1000; do
    for cfg in "0 0" "0 2" "1 2"; do
      set -- $cfg
      LL_TRIDIAG_THREAD=$1 timeout 300 python bench.py --workload c2 --size $size --steps 10 --warmup 2 --cpu-window 0 --tridiag-mode $2 --spmv-reps 5 2>/dev/null \
        | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(json.dumps({'side': $size, 'n': d['config']['n'], 'tridiag_thread': $1, 'tridiag_mode': $2, 'it_per_s': d['value'], 'ms_per_step': d['ms_per_step'], 'host_s_tridiag': d['phases']['host_s_tridiag']}))" >> gpurun_out/r2_small.jsonl
    done
  done
  cat gpurun_out/r2_small.jsonl
fi
if has conv; then
  timeout 600 python tests/convergence_run.py c2 > gpurun_out/r2_conv_c2_defaults.json 2> gpurun_out/r2_conv_c2.err; echo "conv c2 rc=$?"; cut -c1-600 gpurun_out/r2_conv_c2_defaults.json
  LL_TRIDIAG_THREAD=0 timeout 600 python tests/convergence_run.py c2 > gpurun_out/r2_conv_c2_defaults_inline.json 2>> gpurun_out/r2_conv_c2.err; cut -c1-300 gpurun_out/r2_conv_c2_defaults_inline.json
  timeout 600 python tests/convergence_run.py c3 > gpurun_out/r2_conv_c3_defaults.json 2> gpurun_out/r2_conv_c3.err; echo "conv c3 rc=$?"; cut -c1-600 gpurun_out/r2_conv_c3_defaults.json
fi
if has bench; then
  timeout 400 python bench.py --steps 10 --warmup 2 > gpurun_out/r2_bench_c3.json 2> gpurun_out/r2_bench_c3.err; echo "bench rc=$?"; cut -c1-900 gpurun_out/r2_bench_c3.json; tail -3 gpurun_out/r2_bench_c3.err
  timeout 300 python bench.py --workload c5 --steps 10 --warmup 2 > gpurun_out/r2_bench_c5.json 2> gpurun_out/r2_bench_c5.err; echo "bench c5 rc=$?"; python -c "import json; d=json.loads(open('gpurun_out/r2_bench_c5.json').readlines()[-1]); print(d['value'], d['cpu_baseline'])"
  timeout 300 python bench.py --workload c2 --steps 10 --warmup 2 > gpurun_out/r2_bench_c2.json 2> gpurun_out/r2_bench_c2.err; echo "bench c2 rc=$?"; python -c "import json; d=json.loads(open('gpurun_out/r2_bench_c2.json').readlines()[-1]); print(d['value'], d['spmv'], d['cpu_baseline'] and d['cpu_baseline']['value'])"
fi
