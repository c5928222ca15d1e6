#!/bin/bash
# Same-box A/B of one per-context tuning key on bench workloads: interleaved processes, three rounds.
# Usage: tools/ab_tuning_key.sh KEY=VALUE_A KEY=VALUE_B [bench.py options]
A=$1; B=$2; shift 2
for i in 1 2 3 4 5; do
  for t in "$A" "$B"; do
    python3 bench.py "$@" --tuning "$t" --no-other-configs --cpu-window 0 --no-spmv-variants --no-phase-timers --steps 20 --warmup 2 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$t', 'it/s %.1f  ms/step %.4f  spmv_ms %.4f' % (d['value'], d['ms_per_step'], d['spmv']['ms']))"
  done
done
