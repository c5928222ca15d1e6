#!/bin/bash
# Per-rank compute of config 4 on one GPU (stand-in transport) + per-kernel stats at N = 8 and 4.
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/prof_r03
python3 tools/shard_compute_probe.py ${PROBE_RANKS:-1 2 4 8} > gpurun_out/r3_shard_probe.jsonl 2> gpurun_out/r3_shard_probe.err
cat gpurun_out/r3_shard_probe.jsonl
export LL_COMM_PLUGIN=$(pwd)/tests/transport/_build/libll_solo_transport.so
for N in ${PROBE_STATS:-8 4}; do
  d=gpurun_out/prof_r03/shard$N; rm -rf $d
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o shard -- python3 tools/shard_compute_probe.py --child $N > gpurun_out/prof_r03/shard$N.json 2> gpurun_out/prof_r03/shard$N.err
  echo "N=$N rc=$?"
  python3 - $d <<'PY'
import csv,glob,sys
for f in glob.glob(sys.argv[1]+'/**/*kernel_stats.csv', recursive=True):
    rows=list(csv.DictReader(open(f)))
    rows.sort(key=lambda r:-float(r['TotalDurationNs']))
    for r in rows[:14]: print(r['Name'][:70].ljust(72), r['Calls'].rjust(6), ('%.1f'%(float(r['AverageNs'])/1e3)).rjust(8), 'us')
PY
done
