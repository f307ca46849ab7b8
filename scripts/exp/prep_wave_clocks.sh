#!/bin/bash
# EXPERIMENT (round 6): where does k_prep's time go on a SHARD?  (2.1 M reads: 0.08 ms, 16.7 M: 0.22 -- a fixed 0.05 ms.)  The phase-clock
# overlay (scripts/exp/kd_phase_clocks.h) prints, per batch, the wavefronts' average loop and tail times, the longest wavefront and the
# span from the first start to the last end.  Build first:
#   hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -include scripts/exp/kd_phase_clocks.h kindel_amd/csrc/kindel_hip.hip \
#         kindel_amd/csrc/kd_decode.cpp -lz -lpthread -o exp/libkd_phase.so
R=$(cd "$(dirname "$0")/../.." && pwd); cd $R; O=gpurun_out/prep_clocks; mkdir -p $O
for per in 0 8 16 32 64; do
  env="KD_BENCH_LIB=exp/libkd_phase.so"; [ $per -ne 0 ] && env="$env KD_PREP_PER=$per"
  for ranks in 1 8; do
    env $env timeout 600 python scripts/strong_projection.py --config C3 --ranks $ranks --only-rank $((ranks / 2)) --steps 3 --warmup 1 --no-profile --out $O/p.json > /dev/null 2> $O/per${per}_n$ranks.err
    echo "KD_PREP_PER=$per ranks=$ranks: $(grep 'k_prep wavefronts' $O/per${per}_n$ranks.err | tail -1)"
  done
done
