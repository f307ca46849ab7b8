#!/bin/bash
# Exercise bench.py's N > 1 paths (weak: sharded synthesis; strong: one input, work-balanced intervals, reads routed from the
# shared batch; kd_set_shard with shard-local tables; the one fixed-size all-gather) with several ranks sharing the ONE GPU of
# this box over gloo.  NOT a scaling measurement: the strong-scaling FASTA must equal the single-GPU one.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
run() { tag=$1; n=$2; shift 2
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500+RANDOM%200)) bench.py --gpus $n --steps 3 --warmup 1 --backend gloo "$@" > gpurun_out/multirank_$tag.json 2> gpurun_out/multirank_$tag.err
  echo "$tag ranks=$n rc=$?"; grep -iE "error|Traceback" gpurun_out/multirank_$tag.err | head -5
  python - <<PY
import json
try:
    d=json.loads([l for l in open("gpurun_out/multirank_$tag.json") if l.startswith("{")][-1]); o=d.get("other_scaling") or {}
    json.dump(d, open("gpurun_out/multirank_$tag.clean.json", "w"))
    print("  ", d["scaling"], "%.3g ev/s"%d["value"], "%.2f ms"%d["ms_per_step"], d["fasta_sha256"][:12], "| other:", o.get("scaling"), o.get("value") and "%.3g"%o["value"], (o.get("fasta_sha256") or "")[:12])
except Exception as e: print("   failed", e)
PY
}
if [ -n "${MR_QUICK:-}" ]; then      # (two of the four shapes: a short check of the N > 1 path)
  run c3x2 2 --scale 0.2
  run c4x4 4 --config C4 --scale 0.1 --scaling strong --one-scaling
  SINGLES=("C3 0.2" "C4 0.1")
else
  run c3x2 2 --scale 0.2
  run c3x4 4 --scale 0.2 --scaling strong --one-scaling
  run c4x4 4 --config C4 --scale 0.1 --scaling strong --one-scaling
  run c4x8 8 --config C4 --scale 0.05 --scaling strong --one-scaling
  SINGLES=("C3 0.2" "C4 0.1" "C4 0.05")
fi
for c in "${SINGLES[@]}"; do set -- $c
  timeout 300 python bench.py --config $1 --scale $2 --steps 3 --warmup 1 --no-cpu-baseline | python -c "import json,sys; d=json.load(sys.stdin); print('single $1 x$2', '%.3g'%d['value'], d['fasta_sha256'][:12], d['consensus_len'])"
done
