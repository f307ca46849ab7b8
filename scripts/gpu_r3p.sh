#!/bin/bash
# Round 3: C3 bench line (per-kernel table) for a k_prep variant
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; mkdir -p $O
T=${TAG:-r3p}
timeout 600 python bench.py --steps 10 --warmup 3 --e2e-scale 0 ${BENCH_ARGS:-} > $O/${T}_C3.json 2> $O/${T}_C3.err
python - <<PY
import json
d=json.load(open("$O/${T}_C3.json")); print("C3: %.3f ms/step (eager %.3f)"%(d["ms_per_step"], d.get("eager_ms_per_step",0)), {k:round(v["avg_ms"],4) for k,v in d["kernels"].items() if v["avg_ms"]>0.02}, d.get("cpu_baseline",{}).get("bit_exact_vs_gpu"))
PY
