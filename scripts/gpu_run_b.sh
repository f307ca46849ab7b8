#!/bin/bash
# GPU pass B: parity suite, bench C3 with sweep, rocprofv3 kernel-trace stats (csv), C2/C4/C5 quick.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
O=$PWD/gpurun_out
R=$PWD
echo "== pytest gpu"; timeout 1200 python -m pytest tests -m gpu -q --maxfail=8 -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest_gpu.log
echo "== bench C3"; timeout 900 python bench.py --steps 5 --warmup 2 --sweep "${SWEEP:-global:0:0,auto:512:0,auto:2048:0,auto:1024:1024,auto:1024:8192}" > $O/bench_c3.json 2> $O/bench_c3.err; echo "bench rc=$?"; cat $O/bench_c3.json; grep sweep $O/bench_c3.err
echo "== rocprof"; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c3 -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/prof_c3_bench.json 2> $O/prof_c3.err); echo "rocprof rc=$?"
for f in $(find $O/prof_c3 -name "*kernel_stats.csv" | head -1); do head -30 $f; done
echo "== bench C2/C4/C5 quick"
for c in C2 C4 C5; do timeout 400 python bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_$c.json 2> $O/bench_$c.err; python - <<PY
import json
try:
    d=json.load(open("$O/bench_$c.json")); print("$c", "%.3e ev/s"%d["value"], "%.2f ms"%d["ms_per_step"], {k:v["avg_ms"] for k,v in d["kernels"].items() if v["avg_ms"]>0.2})
except Exception as e: print("$c failed", e)
PY
done
