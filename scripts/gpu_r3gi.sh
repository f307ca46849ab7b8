#!/bin/bash
# Round 3: k_gpu_inflate alone, ring 2 / 4 (/ 8 / 16) KiB = 24 / 20 (/ 13 / 8) wavefronts per CU after the far-match fix
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
for v in libgi_2k libgi_4k; do echo "== $v"; timeout 200 python scripts/gpu_inflate_proto.py --lib exp/$v.so --no-verify 2>&1 | grep -E "gpu_ms|rror"; done
