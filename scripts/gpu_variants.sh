#!/bin/bash
# bench lines of library variants with bench arguments: bash scripts/gpu_variants.sh "<lib or ->:<tag>:<bench args>" ...
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=$PWD/gpurun_out; mkdir -p $O
for spec in "$@"; do
  lib=${spec%%:*}; rest=${spec#*:}; tag=${rest%%:*}; args=${rest#*:}
  [ "$lib" = "-" ] && L="" || L="exp/libkd_$lib.so"
  KD_BENCH_LIB=$L timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline $args > $O/var_$tag.json 2> $O/var_$tag.err || tail -2 $O/var_$tag.err
  python - <<PY
import json
try:
    d=json.load(open("$O/var_$tag.json")); k=d["kernels"]
    print("$tag", "%.3f ms"%d["ms_per_step"], d["fasta_sha256"][:8], " ".join("%s=%.3f"%(n[2:],k[n]["avg_ms"]) for n in sorted(k, key=lambda n:-k[n]["avg_ms"]*k[n]["launches_per_step"])[:4]))
except Exception as e: print("$tag failed", e)
PY
done
