#!/bin/bash
# EXPERIMENT (round 6): k_prep's reads per lane (KD_PREP_PER) on the shards of the strong-scaling decomposition of C3 -- one rank of
# every decomposition timed alone (scripts/strong_projection.py --only-rank 1): what should the engine's choice depend on?
R=$(cd "$(dirname "$0")/../.." && pwd); cd $R; O=gpurun_out/prep_sweep; mkdir -p $O
for per in 0 8 16 32 64; do
  env=""; [ $per -ne 0 ] && env="KD_PREP_PER=$per"
  env $env timeout 600 python scripts/strong_projection.py --config C3 --ranks 1,2,4,8 --only-rank 1 --steps 10 --warmup 3 --out $O/per$per.json > /dev/null 2> $O/per$per.err
  python - "$O/per$per.json" "$per" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("KD_PREP_PER=%s" % (sys.argv[2] if sys.argv[2] != "0" else "engine"), " ".join("N=%d: step %.4f k_prep %.4f k_window %.4f |" % (r["n_ranks"], pr["step_ms"], pr["kernels"].get("k_prep", 0), pr["kernels"].get("k_window", 0)) for r in d["rows"] for pr in r["per_rank"]))
PY
done
