#!/bin/bash
# Round 3: kd_step (hipGraph replay): its GPU tests, then the bench line (eager vs graph) on C3 / C2 and the 8-rank shard projection.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "${TESTSEL:-step_graph or cli_two or streamed_ingest or clip_heavy}" > $O/r3h_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/r3h_pytest.log
for c in C3 C2; do
  timeout 600 python bench.py --config $c --steps 10 --warmup 3 --no-cpu-baseline > $O/r3h_$c.json 2> $O/r3h_$c.err; echo "bench $c rc=$?"; tail -3 $O/r3h_$c.err
  python - <<PY
import json
try:
    d=json.load(open("$O/r3h_$c.json")); print("$c: %.3f ms/step (%s) eager %.3f ms; k_window %.4f; value %.3e"%(d["ms_per_step"], d["submission"][:20], d["eager_ms_per_step"], d["kernels"]["k_window"]["avg_ms"], d["value"]))
except Exception as e: print("$c failed", e)
PY
done
