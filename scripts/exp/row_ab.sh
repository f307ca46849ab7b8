#!/bin/bash
# A/B of library builds on ONE box for the N-rank step's tail: a 1/8 shard of a config timed alone with its exchange row registered
# (scripts/strong_projection.py --ranks 8 --only-rank R), alternating:  bash scripts/exp/row_ab.sh <reps> <config> <rank> <lib or "-" for the product> ...
R=$(cd "$(dirname "$0")/../.." && pwd); cd $R
reps=$1; cfg=$2; rank=$3; shift 3
for rep in $(seq 1 $reps); do for lib in "$@"; do
  l=$lib; [ "$lib" = "-" ] && l=""
  env KD_BENCH_LIB=$l python scripts/strong_projection.py --config $cfg --ranks 8 --only-rank $rank --steps 40 --warmup 8 2>/dev/null | python -c "
import json,sys
d=[json.loads(l) for l in sys.stdin if l.startswith('{')][-1]
p=d['rows'][0]['per_rank'][0]
print('$cfg rank $rank %-22s step %.4f ms  kernels %.4f  launches %d  %s' % ('$lib', p['step_ms'], p['kernel_ms'], len(p['kernels']), {k: v for k, v in p['kernels'].items() if k in ('k_cns_emit', 'k_exchange_head', 'k_window', 'k_prep')}))"
done; done
