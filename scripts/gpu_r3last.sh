#!/bin/bash
# Round 3, last GPU seconds: the one-workgroup scans at 1024 threads -- C3 bench line (bit-exactness by fasta_sha256, per-kernel table), then the GPU suite
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=$PWD/gpurun_out; mkdir -p $O
timeout 60 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --e2e-scale 0 > $O/last_c3.json 2> $O/last_c3.err
python - <<PY
import json
d=json.load(open("$O/last_c3.json")); print("C3 %.3f ms (eager %.3f) sha %s"%(d["ms_per_step"], d["eager_ms_per_step"], d["fasta_sha256"][:8]), {k:round(v["avg_ms"],4) for k,v in d["kernels"].items() if "scan" in k or "plan" in k or k=="k_window"})
PY
timeout 100 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $O/pytest_gpu_last.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_gpu_last.log
