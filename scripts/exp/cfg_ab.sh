#!/bin/bash
# A/B of library builds on ONE box over several bench configs, alternating: bash scripts/exp/cfg_ab.sh <reps> "<configs>" <lib or "-" for the product> ...
R=$(cd "$(dirname "$0")/../.." && pwd); cd $R
reps=$1; cfgs=$2; shift 2
for cfg in $cfgs; do for rep in $(seq 1 $reps); do for lib in "$@"; do
  l=$lib; [ "$lib" = "-" ] && l=""
  env KD_BENCH_LIB=$l python bench.py --config $cfg --steps 30 --warmup 8 --no-cpu-baseline --e2e-scale 0 2>/dev/null | python -c "
import json,sys
d=[json.loads(l) for l in sys.stdin if l.startswith('{\"metric')][-1]
print('$cfg %-22s step %.4f ms  kernels %.4f  launches/step %.0f  sha %s' % ('$lib', d['ms_per_step'], d['kernel_ms_per_step'], sum(v['launches_per_step'] for v in d['kernels'].values()), d['fasta_sha256'][:10]))"
done; done; done
