#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; mkdir -p $O
for r in 1 2 4 8 16; do
  KD_SORT_REPS=$r timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-graph --shuffle > $O/r3j_$r.json 2> $O/r3j_$r.err
  python - <<PY
import json
try:
    d=json.load(open("$O/r3j_$r.json")); print("reps $r: %.3f ms"%d["ms_per_step"], {k:v["avg_ms"] for k,v in d["kernels"].items() if v["avg_ms"]>0.05})
except Exception as e: print("reps $r failed", e)
PY
done
