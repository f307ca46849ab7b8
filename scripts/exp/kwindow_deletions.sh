#!/bin/bash
# EXPERIMENT (round 6, TIMING ONLY -- the tables of these builds are wrong): what does k_window's time hang on?  Builds with one part of the
# work DELETED at compile time (-DKD_EXP_DELETE=mask, kd_window.h) against the product, alternating on one box (C3, bench.py):
#   1  the plain reads inside the window add nothing (their base loads stay)       2  ... load nothing (constants instead), adds stay
#   3  neither                                                                       4  no complex walk (clipped / indel reads)
#   8  no flush                                                                     15  all of it: queue, classification, lists, barriers are left
#   for n in 1 2 3 4 8 15; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DKD_EXP_DELETE=$n kindel_amd/csrc/kindel_hip.hip \
#       kindel_amd/csrc/kd_decode.cpp -lz -lpthread -o exp/libkd_del$n.so; done
R=$(cd "$(dirname "$0")/../.." && pwd); cd $R
for rep in 1 2; do for lib in "" exp/libkd_skip2.so exp/libkd_del1.so exp/libkd_del2.so exp/libkd_del3.so exp/libkd_del4.so exp/libkd_del8.so exp/libkd_del15.so; do
  [ -n "$lib" ] && [ ! -f "$lib" ] && continue
  env KD_BENCH_LIB=$lib python bench.py --steps 30 --warmup 8 --no-cpu-baseline --e2e-scale 0 2>/dev/null | python -c "
import json,sys
d=[json.loads(l) for l in sys.stdin if l.startswith('{\"metric')][-1]
print('%-22s step %.4f ms  k_window %.4f ms' % ('${lib:-product}', d['ms_per_step'], d['kernels']['k_window']['avg_ms']))"
done; done
