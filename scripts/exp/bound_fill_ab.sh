#!/bin/bash
# EXPERIMENT RECORD (round 6): k_prep without the boundary table's front / tail fills (status words name the written range, readers
# clamp) against the library before it (exp/libkd_before_prep_pipe.so = the build of profiles/r06_*_bench.json), on ONE box.
R=$(cd "$(dirname "$0")/../.." && pwd); cd $R; O=gpurun_out/bound_fill_ab; mkdir -p $O
for cfgn in C3; do
for cfg in "before:KD_BENCH_LIB=exp/libkd_before_prep_pipe.so" "nofill:KD_X=1" "before:KD_BENCH_LIB=exp/libkd_before_prep_pipe.so" "nofill:KD_X=1"; do
  tag=${cfg%%:*}; env=${cfg#*:}
  env $env timeout 900 python scripts/strong_projection.py --config $cfgn --ranks 1,2,4,8 --steps 10 --warmup 3 --out $O/proj_${cfgn}_$tag.json > /dev/null 2> $O/proj_${cfgn}_$tag.err
  python - "$O/proj_${cfgn}_$tag.json" "$tag" "$cfgn" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("%s %-7s" % (sys.argv[3], sys.argv[2]), " ".join("N=%d: %.4f (k_prep max %.4f) x%.2f |" % (r["n_ranks"], r["projected_step_ms"], max(pr["kernels"].get("k_prep", 0) for pr in r["per_rank"]), r["projected_speedup"]) for r in d["rows"]))
PY
done; done
