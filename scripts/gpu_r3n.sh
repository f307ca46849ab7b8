#!/bin/bash
# Round 3: what k_long_expand's time is made of -- measurement-only builds (exp/libkd_long_*.so: -DKD_EXP_LONG_NOEV / NOROW / NOWALK)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; mkdir -p $O
for v in base "$@"; do
  lib=""; [ "$v" != base ] && lib="$R/exp/libkd_long_$v.so"
  KD_BENCH_LIB=$lib timeout 300 python bench.py --config C5 --steps 5 --warmup 2 --no-cpu-baseline --no-graph --e2e-scale 0 > $O/r3n_$v.json 2> $O/r3n_$v.err
  python - <<PY
import json
d=json.load(open("$O/r3n_$v.json")); print("$v: %.3f ms/step"%d["ms_per_step"], {k:round(x["avg_ms"],4) for k,x in d["kernels"].items() if x["avg_ms"]>0.03})
PY
done
