#!/bin/bash
# Copy the judged summaries of tools/r06_gpu_batch.sh from the scratch directory gpurun_out/ into profiles/ (tracked).
set -u
cd "$(dirname "$0")/.."
for n in default c2 c2lattice c5 c3band c3_steps20 c3_pair_off n1e5 n1e4 c3_window500 c3_window500_pair_off; do cp gpurun_out/r06_final_bench_$n.json profiles/ 2>/dev/null; done
for w in c3 c3band c2; do
  cp gpurun_out/prof_r06/r06_${w}_${w}_kernel_stats.csv profiles/r06_${w}_kernel_stats.csv 2>/dev/null
  cp gpurun_out/prof_r06/r06_bench_${w}_under_rocprof.json profiles/ 2>/dev/null
done
for w in c3 c3band; do cp gpurun_out/prof_r06/r06_${w}_pmc_traffic.json profiles/ 2>/dev/null; done
for w in c3 c2; do cp gpurun_out/r06_convergence_${w}_defaults.json profiles/ 2>/dev/null; done
cp gpurun_out/r06_tl_xcd_probe.txt profiles/ 2>/dev/null
ls -la profiles/r06_* | awk '{print $5, $9}'
cp gpurun_out/r06_shard_compute_probe.txt gpurun_out/r06_convergence_c3_after_seven_processes.json profiles/ 2>/dev/null
cp gpurun_out/r06_shard_probe_banded.txt profiles/r06_shard_compute_probe_banded.txt 2>/dev/null
