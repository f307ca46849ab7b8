#!/bin/bash
# Round 3: k_window_coop on the GPU: parity subset, then bench lines coop vs lane-per-read, window sweep.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; mkdir -p $O
if [ "${SKIP_TESTS:-0}" != 1 ]; then
echo "== parity subset"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "${TESTSEL:-quirk or fixture or tunings or synthetic or unsorted or clip_heavy}" > $O/r3b_pytest.log 2>&1; echo "rc=$?"; tail -3 $O/r3b_pytest.log
fi
echo "== bench coop (default)"; timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --sweep "${SWEEP:-coop:320:0,coop:448:0,coop:576:0,coop:640:0,window:640:0}" > $O/r3b_bench.json 2> $O/r3b_bench.err; echo "rc=$?"
python - <<PY
import json
d=json.load(open("$O/r3b_bench.json")); print("coop default: %.3f ms/step, k_window %.4f ms, frac %.3f"%(d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"]), {k:v["avg_ms"] for k,v in d["kernels"].items()})
for l in open("$O/r3b_bench.err"):
    if l.startswith("{"):
        e=json.loads(l); print(e["sweep"], e["ms_per_step"], e["kernels"].get("k_window"))
PY
