#!/bin/bash
# Round summary on one MI355X: parity suite, smoke, bench C3 (JSON line incl. cpu_baseline), rocprofv3 stats + PMC.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; rm -rf $O/prof_c3 $O/pmc_*; mkdir -p $O
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo "== pytest gpu"; timeout 1200 python -m pytest tests -m gpu -q --maxfail=8 -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_gpu.log
echo "== bench C3"; timeout 900 python bench.py > $O/bench_c3.json 2> $O/bench_c3.err; echo "bench rc=$?"
echo "== rocprof stats"; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c3 -- python $R/bench.py --no-cpu-baseline > $O/prof_c3_bench.json 2> $O/prof_c3.err); echo "rc=$?"
bash scripts/gpu_pmc.sh > $O/pmc_summary.txt 2>&1; grep -E "^[1-4] k_(window|prep|cold)" $O/pmc_summary.txt
echo "== strip mode / shuffled"; bash scripts/gpu_r2.sh lib:none > /dev/null 2>&1; for m in strip; do timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --mode $m > $O/exp_$m.json 2> $O/exp_$m.err; done; timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --shuffle > $O/exp_shuf.json 2> $O/exp_shuf.err
echo "== e2e full C3"; timeout 900 python scripts/e2e_bench.py --scale 1.0 --repeat 3 --out $O/e2e_c3_full.json > /dev/null 2> $O/e2e.err; tail -c 600 $O/e2e_c3_full.json
echo "== multirank (gloo, one GPU)"; bash scripts/gpu_multirank.sh 2>&1 | tail -12
echo "== FETCH_SIZE calibration"; bash scripts/gpu_calib.sh 2>&1 | tail -6
for c in ${CONFIGS:-C2 C4 C5}; do timeout 600 python bench.py --config $c --steps 3 --warmup 1 > $O/bench_$c.json 2> $O/bench_$c.err; done
python - <<PY
import json
for c in ["c3","C2","C4","C5"]:
    try:
        d=json.load(open("$O/bench_%s.json"%c)); print(c, "%.3e ev/s"%d["value"], "%.2f ms"%d["ms_per_step"], "kern %.2f"%d["kernel_ms_per_step"], {k:v["avg_ms"] for k,v in d["kernels"].items() if v["avg_ms"]>0.2}, d.get("cpu_baseline",{}).get("bit_exact_vs_gpu"))
    except Exception as e: print(c, "failed", e)
PY
