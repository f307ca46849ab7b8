#!/bin/bash
# Round 3: unsorted input after the replicated-bin + physical-reorder change; the step test; C3 default for reference.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "step_graph or unsorted or tunings or fixture" > $O/r3i_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/r3i_pytest.log
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --shuffle > $O/exp_shuf.json 2> $O/exp_shuf.err; echo "rc=$?"
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/r3i_sorted.json 2> $O/r3i_sorted.err
python - <<PY
import json
for c in ["exp_shuf","r3i_sorted"]:
    try:
        d=json.load(open("$O/%s.json"%c)); print(c, "%.3f ms (eager %.3f)"%(d["ms_per_step"], d["eager_ms_per_step"]), {k:v["avg_ms"] for k,v in d["kernels"].items() if v["avg_ms"]>0.05})
    except Exception as e: print(c, "failed", e)
PY
