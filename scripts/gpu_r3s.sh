#!/bin/bash
# Round 3: where k_prep's wavefronts spend their clocks (exp/libkd_prepclk.so: HEAD's two-phase loop with s_memtime marks and forced waits),
# and the three-stage software pipeline (product build: 2 reads per step; exp/libkd_pipeU1*.so: 1 read per step)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=$PWD/gpurun_out; mkdir -p $O
KD_BENCH_LIB=exp/libkd_prepclk.so timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-graph --e2e-scale 0 > $O/r3s_clk.json 2> $O/r3s_clk.err
grep "phase clocks" $O/r3s_clk.err | tail -1
bash scripts/gpu_variants.sh "-:pipeU2:--e2e-scale 0" "pipeU1:pipeU1:--e2e-scale 0" "pipeU1o5:pipeU1o5:--e2e-scale 0"
