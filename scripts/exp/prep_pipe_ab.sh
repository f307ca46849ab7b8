#!/bin/bash
# EXPERIMENT RECORD (round 6): k_prep's loop as a load pipeline (kd_prep.h) against the library before it (exp/libkd_before_prep_pipe.so, a
# copy of the previous build), alternating on ONE box: the C3 step, the 1/8 shard, C2 / C4 / C5.
R=$(cd "$(dirname "$0")/../.." && pwd); cd $R; O=gpurun_out/prep_pipe_ab; mkdir -p $O
for rep in 1 2; do
  for cfg in "before:KD_BENCH_LIB=exp/libkd_before_prep_pipe.so" "pipe:KD_X=1"; do
    tag=${cfg%%:*}; env=${cfg#*:}
    env $env timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --e2e-scale 0 > $O/${tag}_$rep.json 2> $O/${tag}_$rep.err
    python - "$O/${tag}_$rep.json" "$tag" "$rep" <<'PY'
import json, sys
d = [json.loads(l) for l in open(sys.argv[1]) if l.startswith('{"metric')][-1]
print("C3 %-7s rep %s  %.4f ms  %.4e ev/s  kernels %.4f  %s  sha %s" % (sys.argv[2], sys.argv[3], d["ms_per_step"], d["value"], d["kernel_ms_per_step"],
      {k: round(v["avg_ms"], 4) for k, v in d["kernels"].items() if k in ("k_window", "k_cns_emit", "k_prep")}, d["fasta_sha256"][:12]))
PY
  done
done
for cfg in "before:KD_BENCH_LIB=exp/libkd_before_prep_pipe.so" "pipe:KD_X=1"; do
  tag=${cfg%%:*}; env=${cfg#*:}
  for c in C2 C4 C5; do
    env $env timeout 300 python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline --e2e-scale 0 > $O/${tag}_$c.json 2> $O/${tag}_$c.err
    python - "$O/${tag}_$c.json" "$tag" "$c" <<'PY'
import json, sys
d = [json.loads(l) for l in open(sys.argv[1]) if l.startswith('{"metric')][-1]
print("%s %-7s %.4f ms  k_prep %s  sha %s" % (sys.argv[3], sys.argv[2], d["ms_per_step"], round(d["kernels"].get("k_prep", {}).get("avg_ms", 0), 4), d["fasta_sha256"][:12]))
PY
  done
  env $env timeout 600 python scripts/strong_projection.py --config C3 --ranks 2,4,8 --only-rank 1 --steps 10 --warmup 3 --out $O/proj_$tag.json > /dev/null 2> $O/proj_$tag.err
  python - "$O/proj_$tag.json" "$tag" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("shards %-7s" % sys.argv[2], " ".join("N=%d: step %.4f k_prep %.4f k_window %.4f |" % (r["n_ranks"], pr["step_ms"], pr["kernels"].get("k_prep", 0), pr["kernels"].get("k_window", 0)) for r in d["rows"] for pr in r["per_rank"]))
PY
done
