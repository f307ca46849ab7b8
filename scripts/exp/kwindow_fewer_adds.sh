#!/bin/bash
# EXPERIMENT (round 6, TIMING ONLY -- the tables of these builds are wrong): what could a scheme that removes ds_add instructions from
# k_window's hot path buy AT MOST?  (The review's idea: two bases of one read that fall into one site pair with equal channel as ONE
# ds_add of 0x10001 -- at best 15 % fewer lane-adds, at the price of a compare, a select and an exec mask per pair.)  Builds with the
# last 1 / 2 bases of every dword of a plain read simply not added (-DKD_EXP_SKIP_BASES=1 / 2: 12.5 / 25 % fewer ds_add AND their
# address arithmetic, for free) against the product, alternating on one box:
#   for n in 1 2; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DKD_EXP_SKIP_BASES=$n kindel_amd/csrc/kindel_hip.hip \
#       kindel_amd/csrc/kd_decode.cpp -lz -lpthread -o exp/libkd_skip$n.so; done
R=$(cd "$(dirname "$0")/../.." && pwd); cd $R
for rep in 1 2 3; do for lib in "" exp/libkd_skip1.so exp/libkd_skip2.so; do
  env KD_BENCH_LIB=$lib python bench.py --steps 30 --warmup 8 --no-cpu-baseline --e2e-scale 0 2>/dev/null | python -c "
import json,sys
d=[json.loads(l) for l in sys.stdin if l.startswith('{\"metric')][-1]
print('%-22s step %.4f ms  k_window %.4f ms' % ('${lib:-product}', d['ms_per_step'], d['kernels']['k_window']['avg_ms']))"
done; done
