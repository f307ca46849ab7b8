#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
for tl in 256 512 1024 2048; do KD_TILE=$tl timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --sweep "auto:192:0,auto:320:0,auto:448:0" > gpurun_out/tile_$tl.json 2> gpurun_out/tile_$tl.err; python - <<PY
import json; d=json.load(open("gpurun_out/tile_$tl.json")); print("TILE=$tl W=256", "%.2f ms"%d["ms_per_step"], "k_window", d["kernels"]["k_window"]["avg_ms"])
for l in open("gpurun_out/tile_$tl.err"):
    if l.startswith('{"sweep'):
        s=json.loads(l); print("   ", s["sweep"], s["ms_per_step"], s["kernels"]["k_window"])
PY
done
