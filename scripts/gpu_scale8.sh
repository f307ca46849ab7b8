#!/bin/bash
# For a driver that HAS an 8-GPU MI355X node (gpurun boxes have one GPU): the scaling curve of BASELINE.json's metric, strong (one
# C3 input cut into N work-balanced position intervals: north_star's ">= 6 x at 8 GPUs") and weak (one C3-sized interval per rank),
# N = 1 2 4 8, launched exactly as the round driver launches bench.py, with RCCL's own log proving the rank count of every run.
#   bash scripts/gpu_scale8.sh [OUTDIR]      -> OUTDIR/scale_{strong,weak}_N.json, OUTDIR/scale_summary.txt
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
O=${1:-gpurun_out/scale8}; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0 NCCL_DEBUG=INFO NCCL_DEBUG_SUBSYS=INIT
NG=$(python -c "import torch; print(torch.cuda.device_count())")
: > $O/scale_summary.txt
for sc in strong weak; do
  for n in 1 2 4 8; do
    [ $n -gt $NG ] && { echo "$sc N=$n skipped: $NG GPU(s) visible" | tee -a $O/scale_summary.txt; continue; }
    port=$((29600 + n))
    if [ $n -eq 1 ]; then cmd="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --e2e-scale 0"
    else cmd="python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $port bench.py --gpus $n --steps 20 --warmup 5 --scaling $sc --one-scaling --no-cpu-baseline --e2e-scale 0"; fi
    timeout 1200 $cmd > $O/scale_${sc}_$n.json 2> $O/scale_${sc}_$n.err; rc=$?
    ranks=$(grep -a -o "nranks [0-9]*" $O/scale_${sc}_$n.err $O/scale_${sc}_$n.json | sort -u | tr '\n' ' ')
    python - "$O/scale_${sc}_$n.json" "$sc" "$n" "$rc" "$ranks" <<'PY' | tee -a $O/scale_summary.txt
import json, sys
path, sc, n, rc, ranks = sys.argv[1:6]
try:
    d = [json.loads(l) for l in open(path) if l.startswith('{"metric')][-1]
    print("%s N=%s rc=%s  %.4e events/s  %.4f ms/step  scaling=%s  fasta %s  RCCL: %s" % (sc, n, rc, d["value"], d["ms_per_step"], d.get("scaling"), d.get("fasta_sha256", "")[:12], ranks or "-"))
except Exception as e:
    print("%s N=%s rc=%s  no bench line (%s)" % (sc, n, rc, e))
PY
  done
done
