#!/bin/bash
# A/B of library builds on ONE box, alternating: bash scripts/exp/lib_ab.sh <reps> <config> <lib or "-" for the product> ...
R=$(cd "$(dirname "$0")/../.." && pwd); cd $R
reps=$1; cfg=$2; shift 2
for rep in $(seq 1 $reps); do for lib in "$@"; do
  l=$lib; [ "$lib" = "-" ] && l=""
  env KD_BENCH_LIB=$l python bench.py --config $cfg --steps 30 --warmup 8 --no-cpu-baseline --e2e-scale 0 2>/dev/null | python -c "
import json,sys
d=[json.loads(l) for l in sys.stdin if l.startswith('{\"metric')][-1]
print('$cfg %-22s step %.4f ms  k_window %.4f ms  sha %s' % ('$lib', d['ms_per_step'], d['kernels']['k_window']['avg_ms'], d['fasta_sha256'][:10]))"
done; done
