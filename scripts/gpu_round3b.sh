#!/bin/bash
# Round 3, final summary on one MI355X: smoke, the whole GPU parity suite, bench C3 (JSON line incl. cpu_baseline + live e2e with the
# device-side ingest), rocprofv3 stats + PMC of the same command, the other configs, shuffled input, strong-scaling projections,
# full-size e2e (Phred / absent qualities: host decoder whole-file + streamed, device-side ingest).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; rm -rf $O/prof_c3 $O/pmc_*; mkdir -p $O
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo "== pytest gpu"; timeout 1200 python -m pytest tests -m gpu -q --maxfail=8 -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
echo "== bench C3"; timeout 600 python bench.py > $O/bench_c3.json 2> $O/bench_c3.err; echo "bench rc=$?"
echo "== rocprof stats"; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c3 -- python $R/bench.py --no-cpu-baseline > $O/prof_c3_bench.json 2> $O/prof_c3.err); echo "rc=$?"
bash scripts/gpu_pmc.sh > $O/pmc_summary.txt 2>&1; grep -E "^[1-4] k_(window|prep|cold)" $O/pmc_summary.txt
for c in C2 C4 C5; do timeout 300 python bench.py --config $c --steps 10 --warmup 3 --e2e-scale 0 > $O/bench_$c.json 2> $O/bench_$c.err; done
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --shuffle > $O/exp_shuf.json 2> $O/exp_shuf.err
echo "== projections"; for c in C3 C4; do timeout 400 python scripts/strong_projection.py --config $c --out $O/r03_strong_scaling_projection_$c.json > $O/proj_$c.log 2>&1; tail -2 $O/proj_$c.log; done
echo "== e2e full C3"; timeout 600 python scripts/e2e_bench.py --scale 1.0 --repeat 2 --qual phred --out $O/e2e_c3_full_phred.json > /dev/null 2> $O/e2e_phred.err; tail -c 700 $O/e2e_c3_full_phred.json; echo
timeout 600 python scripts/e2e_bench.py --scale 1.0 --repeat 2 --check --out $O/e2e_c3_full.json > /dev/null 2> $O/e2e.err; tail -c 700 $O/e2e_c3_full.json; echo
python - <<PY
import json
for c in ["c3","C2","C4","C5"]:
    try:
        d=json.load(open("$O/bench_%s.json"%c)); print(c, "%.3e ev/s"%d["value"], "%.3f ms (eager %.3f)"%(d["ms_per_step"], d["eager_ms_per_step"]), "kern %.2f"%d["kernel_ms_per_step"], {k:v["avg_ms"] for k,v in d["kernels"].items() if v["avg_ms"]>0.1}, d.get("cpu_baseline",{}).get("bit_exact_vs_gpu"), "frac", d.get("roofline",{}).get("frac"))
    except Exception as e: print(c, "failed", e)
try:
    d=json.load(open("$O/exp_shuf.json")); print("shuf %.3f ms"%d["ms_per_step"], {k:v["avg_ms"] for k,v in d["kernels"].items() if v["avg_ms"]>0.1})
    d=json.load(open("$O/bench_c3.json")); print("e2e live", json.dumps(d.get("e2e"))[:900])
except Exception as e: print("failed", e)
PY
