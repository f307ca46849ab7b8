#!/bin/bash
# quick GPU pass: parity suite + bench C3 sweep (+ optional extra configs)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
O=$PWD/gpurun_out
echo "== pytest gpu"; timeout 1200 python -m pytest tests -m gpu -q --maxfail=8 -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.log
echo "== bench C3"; timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --sweep "${SWEEP:-auto:512:0,auto:2048:0,auto:1024:1024,auto:1024:8192}" > $O/bench_c3.json 2> $O/bench_c3.err; echo "bench rc=$?"
python - <<PY
import json
d=json.load(open("$O/bench_c3.json"))
print("C3 %.3e ev/s  %.2f ms/step  kernels %.2f ms"%(d['value'], d['ms_per_step'], d['kernel_ms_per_step']))
print({k:v['avg_ms'] for k,v in d['kernels'].items()})
for l in open("$O/bench_c3.err"):
    if l.startswith('{"sweep'):
        s=json.loads(l); print(s['sweep'], s['ms_per_step'], s['items'], {k:v for k,v in s['kernels'].items() if v>0.3})
PY
for c in ${CONFIGS:-C2 C4 C5}; do timeout 400 python bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_$c.json 2> $O/bench_$c.err; python - <<PY
import json
try:
    d=json.load(open("$O/bench_$c.json")); print("$c", "%.3e ev/s"%d["value"], "%.2f ms"%d["ms_per_step"], {k:v["avg_ms"] for k,v in d["kernels"].items() if v["avg_ms"]>0.2})
except Exception as e: print("$c failed", e)
PY
done
