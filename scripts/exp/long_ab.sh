#!/bin/bash
# A/B of library builds on the long-read workload (C5) on ONE box, alternating: bash scripts/exp/long_ab.sh <reps> <lib or "-" for the product> ...
R=$(cd "$(dirname "$0")/../.." && pwd); cd $R
reps=$1; shift 1
for rep in $(seq 1 $reps); do for lib in "$@"; do
  l=$lib; [ "$lib" = "-" ] && l=""
  env KD_BENCH_LIB=$l python bench.py --config C5 --steps 30 --warmup 8 --no-cpu-baseline --e2e-scale 0 2>/dev/null | python -c "
import json,sys
d=[json.loads(l) for l in sys.stdin if l.startswith('{\"metric')][-1]
k=d['kernels']
print('C5 %-22s step %.4f ms  kernels %.4f  %s  sha %s' % ('$lib', d['ms_per_step'], d['kernel_ms_per_step'], ' '.join('%s %.4f' % (n[2:], k[n]['avg_ms']) for n in ('k_prep_long','k_long_expand','k_long_order','k_long_reduce','k_window_rows','k_sort_small','k_sort_count','k_sort_scan','k_sort_scatter') if n in k), d['fasta_sha256'][:10]))"
done; done
