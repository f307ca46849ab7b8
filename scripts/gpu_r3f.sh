#!/bin/bash
# Round 3: the whole GPU parity suite + strong-scaling projection (one GPU) + default bench line.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; mkdir -p $O
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo "== pytest gpu"; timeout 2400 python -m pytest tests -m gpu -q --maxfail=8 -p no:cacheprovider --durations=8 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -14 $O/pytest_gpu.log
echo "== strong-scaling projection"; timeout 900 python scripts/strong_projection.py --config C3 --out $O/r03_strong_scaling_projection_C3.json > $O/proj_c3.log 2>&1; echo "rc=$?"; tail -c 400 $O/proj_c3.log
timeout 900 python scripts/strong_projection.py --config C4 --out $O/r03_strong_scaling_projection_C4.json > $O/proj_c4.log 2>&1; echo "rc=$?"; tail -c 400 $O/proj_c4.log
echo "== bench"; timeout 900 python bench.py > $O/bench_c3.json 2> $O/bench_c3.err; echo "rc=$?"; tail -c 1500 $O/bench_c3.json
