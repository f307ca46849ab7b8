#!/bin/bash
# Round 3: k_gpu_inflate variants (literal runs with the next look-up in flight; 4 KiB ring = 20 instead of 13 wavefronts per CU), inflate alone
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
for v in base runs base4k runs4k; do echo "== $v"; timeout 200 python scripts/gpu_inflate_proto.py --lib exp/libgi_$v.so --no-verify 2>&1 | grep -E "gpu_ms|Error|error" ; done
