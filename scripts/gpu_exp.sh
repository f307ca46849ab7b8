#!/bin/bash
# scratch driver for timing experiments on the GPU box (bench lines + optional pytest)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
run() { tag=$1; shift; python bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/exp_$tag.json 2> gpurun_out/exp_$tag.err || tail -3 gpurun_out/exp_$tag.err; grep "phase clocks" gpurun_out/exp_$tag.err | tail -1; python - <<PY
import json; d=json.load(open("gpurun_out/exp_$tag.json")); k=d["kernels"]
print("$tag", "%.2f ms"%d["ms_per_step"], "ksum %.2f"%d["kernel_ms_per_step"], "events %.3g"%d["config"]["aligned_events"], d["fasta_sha256"][:8], " ".join("%s=%.3f"%(n[2:],k[n]["avg_ms"]) for n in sorted(k, key=lambda n:-k[n]["avg_ms"])[:7]))
PY
}
for a in "$@"; do
  case $a in
    base) run base;;
    c2) run c2 --config C2;;
    c4) run c4 --config C4;;
    c5) run c5 --config C5;;
    c5w) for w in 256 1024; do run c5_w$w --config C5 --window $w; done;;
    c5s) for sl in 64 128; do run c5_s$sl --config C5 --slice $sl; done;;
    shuf) run shuf --shuffle;;
    plain) run plain --synth clip_p=0.0 --synth indel_p=0.0 --synth planted=0;;
    phase) KD_BENCH_LIB=exp/libkd_phase.so run phase;;
    phaseplain) KD_BENCH_LIB=exp/libkd_phase.so run phaseplain --synth clip_p=0.0 --synth indel_p=0.0 --synth planted=0;;   # hipcc -DKD_PHASE_CLOCKS build in exp/
    tests) timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3;;
  esac
done
