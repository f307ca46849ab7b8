#!/bin/bash
# Exercise bench.py's N > 1 path (sharded synthesis, kd_set_shard, all-gather stitch) with several ranks sharing
# the one GPU of this box over gloo.  Not a scaling measurement.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
for n in 2 4; do
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500+n)) bench.py --gpus $n --steps 3 --warmup 1 --backend gloo --scale 0.2 > gpurun_out/multirank_$n.json 2> gpurun_out/multirank_$n.err
  echo "ranks=$n rc=$?"; tail -c 900 gpurun_out/multirank_$n.json; echo; grep -iE "error|Traceback" gpurun_out/multirank_$n.err | head -5
done
timeout 300 python bench.py --scale 0.2 --steps 3 --warmup 1 --no-cpu-baseline | python -c "import json,sys; d=json.load(sys.stdin); print('single', d['value'], d['fasta_sha256'][:16], d['consensus_len'])"
