#!/bin/bash
# Round 3: projections (C3, C4), other configs' bench lines, shuffled input, timeline, rocprofv3 stats + PMC for profiles/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; mkdir -p $O
echo "== projections"; for c in C3 C4; do timeout 900 python scripts/strong_projection.py --config $c --out $O/r03_strong_scaling_projection_$c.json > $O/proj_$c.log 2>&1; echo "$c rc=$?"; done
python - <<PY
import json
for c in ("C3","C4"):
    try:
        p=json.load(open("$O/r03_strong_scaling_projection_%s.json"%c)); print(c, [(r["n_ranks"], r["projected_step_ms"], r["projected_speedup"]) for r in p["rows"]], p["fixed_ms_estimate"])
    except Exception as e: print(c, "failed", e)
PY
echo "== other configs"; for c in C2 C4 C5; do timeout 600 python bench.py --config $c --steps 5 --warmup 2 --e2e-scale 0 > $O/bench_$c.json 2> $O/bench_$c.err; done
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --shuffle > $O/bench_c3_shuffled.json 2> $O/bench_c3_shuffled.err
python - <<PY
import json
for c in ["C2","C4","C5","c3_shuffled"]:
    try:
        d=json.load(open("$O/bench_%s.json"%c)); print(c, "%.3e ev/s"%d["value"], "%.3f ms"%d["ms_per_step"], "kern %.3f"%d["kernel_ms_per_step"], {k:v["avg_ms"] for k,v in d["kernels"].items() if v["avg_ms"]>0.1}, d.get("cpu_baseline",{}).get("bit_exact_vs_gpu"))
    except Exception as e: print(c, "failed", e)
PY
echo "== rocprof stats"; rm -rf $O/prof_c3; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c3 -- python $R/bench.py --no-cpu-baseline > $O/prof_c3_bench.json 2> $O/prof_c3.err); echo "rc=$?"
find $O/prof_c3 -name "*kernel_stats.csv" | head -2
bash scripts/gpu_pmc.sh > $O/pmc_summary.txt 2>&1; grep -E "^[1-4] k_(window|prep|cold)" $O/pmc_summary.txt
