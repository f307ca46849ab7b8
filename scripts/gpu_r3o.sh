#!/bin/bash
# Round 3: C5 bench line + where the wavefronts of the ROW pass (k_window<true>) spend their clocks (exp/libkd_phase_rows.so: -DKD_PHASE_CLOCKS)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout 300 python bench.py --config C5 --steps 10 --warmup 3 --e2e-scale 0 > $O/r3o_C5.json 2> $O/r3o_C5.err
python - <<PY
import json
d=json.load(open("$O/r3o_C5.json")); print("C5: %.3f ms/step (eager %.3f)"%(d["ms_per_step"], d.get("eager_ms_per_step",0)), {k:round(v["avg_ms"],4) for k,v in d["kernels"].items() if v["avg_ms"]>0.015}, d.get("cpu_baseline",{}).get("bit_exact_vs_gpu"))
PY
KD_BENCH_LIB=$R/exp/libkd_phase_rows.so timeout 300 python bench.py --config C5 --steps 1 --warmup 0 --no-cpu-baseline --no-graph --e2e-scale 0 > $O/r3o_phase.json 2> $O/r3o_phase.err
grep "phase clocks" $O/r3o_phase.err | tail -2
