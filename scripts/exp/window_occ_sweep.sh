#!/bin/bash
# EXPERIMENT (round 6): k_window at FOUR workgroups per CU -- 128 registers instead of 96 (-DKD_WINDOW_OCC=4) and windows of 512 - 640 sites --
# with what the fifth workgroup's registers pay for: kd_walk_short two chunks ahead (B), the next row's offsets one row ahead (R), both (BR).
# Libraries: exp/libkd_occ4[_B|_R|_BR].so (built from patched copies of kindel_amd/csrc, see profiles/r06_kwindow_deletions.txt).
R=$(cd "$(dirname "$0")/../.." && pwd); cd $R
for lib in "" exp/libkd_occ4.so exp/libkd_occ4_B.so exp/libkd_occ4_R.so exp/libkd_occ4_BR.so; do
  [ -n "$lib" ] && [ ! -f "$lib" ] && continue
  echo "== lib ${lib:-product} =="
  env KD_BENCH_LIB=$lib python bench.py --steps 20 --warmup 5 --no-cpu-baseline --e2e-scale 0 --sweep window:448:0,window:512:0,window:576:0,window:640:0,window:448:0,window:512:0,window:576:0 2>&1 >/dev/null | grep '"sweep"' | python -c "
import json,sys
print(' | '.join('%s %.4f' % (json.loads(l)['sweep'].split(':')[1], json.loads(l)['ms_per_step']) for l in sys.stdin))"
done
