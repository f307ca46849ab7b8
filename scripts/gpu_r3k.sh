#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "unsorted or tunings or fixture or shuffl" 2>&1 | tail -3
run() { # name, env..., args
  name=$1; shift
  env "$@" > /dev/null 2>&1
}
for v in lds global; do
  if [ $v = global ]; then export KD_SORT_GLOBAL=1; else unset KD_SORT_GLOBAL; fi
  timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --shuffle > $O/r3k_shuf_$v.json 2> $O/r3k_shuf_$v.err
  python - <<PY
import json
try:
    d=json.load(open("$O/r3k_shuf_$v.json")); print("shuffled $v: %.3f ms (eager %s)"%(d["ms_per_step"], d.get("eager_ms_per_step")), {k:v["avg_ms"] for k,v in d["kernels"].items() if v["avg_ms"]>0.02})
except Exception as e: print("shuffled $v failed", e)
PY
done
unset KD_SORT_GLOBAL
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/r3k_sorted.json 2> $O/r3k_sorted.err
python - <<PY
import json
d=json.load(open("$O/r3k_sorted.json")); print("sorted: %.3f ms (eager %s)"%(d["ms_per_step"], d.get("eager_ms_per_step")), {k:v["avg_ms"] for k,v in d["kernels"].items() if v["avg_ms"]>0.02})
PY
