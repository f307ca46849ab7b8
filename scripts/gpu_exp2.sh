#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
run() { tag=$1; shift; python bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/exp_$tag.json 2> gpurun_out/exp_$tag.err || tail -3 gpurun_out/exp_$tag.err; python - <<PY
import json; d=json.load(open("gpurun_out/exp_$tag.json")); k=d["kernels"]
print("$tag", "%.2f ms"%d["ms_per_step"], "ksum %.2f"%d["kernel_ms_per_step"], "events %.3g"%d["config"]["aligned_events"], d["fasta_sha256"][:8], " ".join("%s=%.3f"%(n[2:],k[n]["avg_ms"]) for n in ("k_window","k_prep","k_cold_lane","k_ins_insert","k_cns_count") if n in k))
PY
}
run base
run c4 --config C4
