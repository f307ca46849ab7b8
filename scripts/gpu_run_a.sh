#!/bin/bash
# First GPU pass: smoke, parity suite, bench (C3) with tuning sweep, rocprofv3 kernel trace.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9|Compute Unit|Max Waves|LDS|Wavefront Size" | head -20 > $O/rocminfo.txt
echo "== smoke" ; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 $O/smoke.log
echo "== pytest gpu"; timeout 1200 python -m pytest tests -m gpu -q --maxfail=8 -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -25 $O/pytest_gpu.log
echo "== bench C3"; timeout 900 python bench.py --steps 5 --warmup 2 --sweep "global:0:0,auto:1024:0,auto:4096:0,auto:2048:1024,auto:2048:8192,auto:512:0" > $O/bench_c3.json 2> $O/bench_c3.err; echo "bench rc=$?"; cat $O/bench_c3.json; tail -12 $O/bench_c3.err
echo "== rocprof"; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OLDPWD/$O/prof_c3 -- python $OLDPWD/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OLDPWD/$O/prof_c3_bench.json 2> $OLDPWD/$O/prof_c3.err); echo "rocprof rc=$?"
find $O/prof_c3 -name "*stats*" | head; for f in $(find $O/prof_c3 -name "*kernel_stats.csv" | head -1); do head -25 $f; done
echo "== bench C2/C5 quick"; timeout 300 python bench.py --config C2 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err; cat $O/bench_c2.json
timeout 300 python bench.py --config C5 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err; cat $O/bench_c5.json; tail -3 $O/bench_c5.err
