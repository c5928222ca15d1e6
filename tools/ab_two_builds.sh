#!/bin/bash
# same-box A/B of two builds of the library (lambda-lanczos_amd/lib_base vs lib): interleaved processes
export LL_COMM_PLUGIN=$(pwd)/tests/transport/_build/libll_solo_transport.so
for i in 1 2 3; do
  for which in base new; do
    if [ $which = base ]; then export LL_LIB_PATH=$(pwd)/lambda-lanczos_amd/lib_base/liblanczos_hip.so; else unset LL_LIB_PATH; fi
    python3 tools/shard_compute_probe.py --child ${AB_RANKS:-8} 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); p=d['per_iteration_us']; print('$which', 'spmv_ms %.4f operator %.1f gs %.1f wall %.1f'%(d['spmv_ms_incl_local_copies'], p['operator_incl_local_copies'], p['gram_schmidt'], p['wall']))"
  done
done
