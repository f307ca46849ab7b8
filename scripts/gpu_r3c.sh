#!/bin/bash
# Round 3: where does k_window_coop spend its time?  Measurement-only builds (exp/libkd_coop_*.so) of the same bench command.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; mkdir -p $O
for v in ${VARIANTS:-base NOWALK NOPLAIN NOCPLX NOLOAD2}; do
  if [ $v = base ]; then unset KD_BENCH_LIB; else export KD_BENCH_LIB=$R/exp/libkd_coop_$v.so; fi
  timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline ${BENCH_ARGS:-} > $O/r3c_$v.json 2> $O/r3c_$v.err
  python - <<PY
import json
try:
    d=json.load(open("$O/r3c_$v.json")); print("$v: %.3f ms/step, k_window %.4f ms"%(d["ms_per_step"], d["kernels"]["k_window"]["avg_ms"]))
except Exception as e: print("$v failed", e)
PY
done
