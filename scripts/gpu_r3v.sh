#!/bin/bash
# Round 3: the tree after the k_prep rewrite on a fresh box -- GPU parity suite, then bench lines C3 (default), C2, C4, C5, shuffled C3
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; mkdir -p $O
echo "== pytest gpu"; timeout 1200 python -m pytest tests -m gpu -q --maxfail=8 -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
timeout 600 python bench.py > $O/bench_c3.json 2> $O/bench_c3.err; echo "bench rc=$?"
for c in C2 C4 C5; do timeout 600 python bench.py --config $c --steps 10 --warmup 3 --e2e-scale 0 > $O/bench_$c.json 2> $O/bench_$c.err; done
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --shuffle > $O/exp_shuf.json 2> $O/exp_shuf.err
python - <<PY
import json
for c in ["bench_c3","bench_C2","bench_C4","bench_C5","exp_shuf"]:
    try:
        d=json.load(open("$O/%s.json"%c)); print(c, "%.3e ev/s"%d["value"], "%.3f ms (eager %.3f)"%(d["ms_per_step"], d.get("eager_ms_per_step",0)), {k:round(v["avg_ms"],4) for k,v in d["kernels"].items() if v["avg_ms"]>0.02}, d.get("cpu_baseline",{}).get("bit_exact_vs_gpu"))
    except Exception as e: print(c, "failed", e)
PY
