#!/bin/bash
# Round 3: k_window with channel-major LDS rows + residue-ordered lists: parity, bench, sweep, the unsorted-list build.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; mkdir -p $O
if [ "${SKIP_TESTS:-0}" != 1 ]; then
echo "== parity subset"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "${TESTSEL:-quirk or fixture or tunings or synthetic or unsorted or clip_heavy}" > $O/r3d_pytest.log 2>&1; echo "rc=$?"; tail -3 $O/r3d_pytest.log
fi
for v in base ${VARIANTS:-nosort}; do
  if [ $v = base ]; then unset KD_BENCH_LIB; else export KD_BENCH_LIB=$R/exp/libkd_$v.so; fi
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline ${BENCH_ARGS:-} --sweep "${SWEEP:-window:448:0,window:576:0,window:640:0,window:832:0}" > $O/r3d_$v.json 2> $O/r3d_$v.err
  python - <<PY
import json
try:
    d=json.load(open("$O/r3d_$v.json")); print("$v: %.3f ms/step, k_window %.4f ms"%(d["ms_per_step"], d["kernels"]["k_window"]["avg_ms"]), {k:x["avg_ms"] for k,x in d["kernels"].items() if x["avg_ms"]>0.05})
    for l in open("$O/r3d_$v.err"):
        if l.startswith("{"):
            e=json.loads(l); print("   ", e["sweep"], e["ms_per_step"], e["kernels"].get("k_window"))
except Exception as e: print("$v failed", e)
PY
done
