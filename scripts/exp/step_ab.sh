#!/bin/bash
# EXPERIMENT RECORD (round 6): the C3 step with round 5's submission (k_cold_lane a launch of its own, the FASTA copied behind the
# consensus kernel) against this round's defaults (cold records in k_window's launch, the consensus kernel writing the pinned buffer),
# alternating on ONE box -- box-to-box spread is larger than the difference.
R=$(cd "$(dirname "$0")/../.." && pwd); cd $R; O=gpurun_out/step_ab; mkdir -p $O
for rep in 1 2 3; do
  for cfg in "old:KD_COLD_TAIL=0 KD_ZERO_COPY=0" "tail_only:KD_COLD_TAIL=1 KD_ZERO_COPY=0" "new:KD_COLD_TAIL=1 KD_ZERO_COPY=1"; do
    tag=${cfg%%:*}; env=${cfg#*:}
    env $env timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --e2e-scale 0 > $O/${tag}_$rep.json 2> $O/${tag}_$rep.err
    python - "$O/${tag}_$rep.json" "$tag" "$rep" <<'PY'
import json, sys
d = [json.loads(l) for l in open(sys.argv[1]) if l.startswith('{"metric')][-1]
print("%-9s rep %s  %.4f ms  %.4e ev/s  kernels %.4f  %s  sha %s" % (sys.argv[2], sys.argv[3], d["ms_per_step"], d["value"], d["kernel_ms_per_step"],
      {k: round(v["avg_ms"], 4) for k, v in d["kernels"].items() if k in ("k_window", "k_cold_lane", "k_cns_emit", "k_prep")}, d["fasta_sha256"][:12]))
PY
  done
done
