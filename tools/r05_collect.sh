#!/bin/bash
# Copy the judged summaries of tools/r05_gpu_batch.sh from the scratch directory gpurun_out/ into profiles/ (tracked).
set -u
cd "$(dirname "$0")/.."
for n in default c2 c2lattice c5 c3band c3_steps20 c3_pair_off; do cp gpurun_out/r05_final_bench_$n.json profiles/ 2>/dev/null; done
for w in c3 c3band c2; do
  cp gpurun_out/prof_r05/r05_${w}_${w}_kernel_stats.csv profiles/r05_${w}_kernel_stats.csv 2>/dev/null
  cp gpurun_out/prof_r05/r05_bench_${w}_under_rocprof.json profiles/ 2>/dev/null
done
for w in c3 c3band; do cp gpurun_out/prof_r05/r05_${w}_pmc_traffic.json profiles/ 2>/dev/null; done
for w in c3 c2; do cp gpurun_out/r05_convergence_${w}_defaults.json profiles/ 2>/dev/null; done
cp gpurun_out/r05_tl_xcd_probe.txt profiles/ 2>/dev/null
ls -la profiles/r05_* | awk '{print $5, $9}'
