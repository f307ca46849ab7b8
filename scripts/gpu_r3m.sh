#!/bin/bash
# Round 3: long reads as ROWS (kd_long.h): parity of the long-read tests on the GPU, then the C5 bench line (with the full-size
# bit-exactness check against the oracle) and its per-kernel table.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_fuzz.py -x -q -m gpu -k "long or C5 or shards" > $O/r3m_tests.log 2>&1
tail -5 $O/r3m_tests.log
timeout 600 python bench.py --config C5 --steps 10 --warmup 3 --e2e-scale 0 > $O/r3m_C5.json 2> $O/r3m_C5.err
tail -3 $O/r3m_C5.err
python - <<PY
import json
d=json.load(open("$O/r3m_C5.json")); print("C5: %.3f ms/step (eager %.3f)"%(d["ms_per_step"], d.get("eager_ms_per_step",0)), {k:round(v["avg_ms"],4) for k,v in d["kernels"].items() if v["avg_ms"]>0.015})
print("bit exact:", d.get("cpu_baseline",{}).get("bit_exact_vs_gpu"))
PY
