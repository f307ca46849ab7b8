#!/bin/bash
# Round 3: the counting sort without global atomics (count rows -> column scan -> scatter): GPU tests that sort, then shuffled C3 in both layouts
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=$PWD/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "unsorted or shuffl or order or hxb2 or fixture" > $O/pytest_sort.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_sort.log
bash scripts/gpu_variants.sh "-:shuf_records:--e2e-scale 0 --shuffle" "-:shuf_index:--e2e-scale 0 --shuffle index"
