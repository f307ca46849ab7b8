#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
for d in 0 1 2 3; do KD_DEBUG=$d timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/dbg_$d.json 2>/dev/null; python - <<PY
import json; d=json.load(open("gpurun_out/dbg_$d.json")); print("KD_DEBUG=$d", "%.2f ms"%d["ms_per_step"], {k:v["avg_ms"] for k,v in d["kernels"].items() if v["avg_ms"]>0.3})
PY
done
