#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
run() { python bench.py --steps 3 --warmup 1 --no-cpu-baseline --sweep "auto:448:0,auto:896:0" > gpurun_out/exp.json 2> gpurun_out/exp.err; python - <<PY
import json; d=json.load(open("gpurun_out/exp.json")); print("$1", "W=640 %.2f ms"%d["ms_per_step"], "k_window", d["kernels"]["k_window"]["avg_ms"], end=" | ")
for l in open("gpurun_out/exp.err"):
    if l.startswith('{"sweep'):
        s=json.loads(l); print(s["sweep"], s["kernels"]["k_window"], end=" ")
print()
PY
}
KD_TILE=1024 run "tile1024"
KD_TILE=512 run "tile512"
KD_TILE=1024 KD_WHPAD=1 run "tile1024+pad"
KD_TILE=512 KD_WHPAD=1 run "tile512+pad"
KD_TILE=256 KD_WHPAD=1 run "tile256+pad"
