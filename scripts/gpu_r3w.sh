#!/bin/bash
# Round 3: unsorted input as a decoder hands it over (--shuffle records: payload in record order) vs the index-only permutation of rounds 1 - 2
cd "${GRAFT_REPO_ROOT:-/root/repo}"
bash scripts/gpu_variants.sh "-:shuf_records:--e2e-scale 0 --shuffle" "-:shuf_index:--e2e-scale 0 --shuffle index" "${EXTRA_LIB:--}:shuf_records_x:--e2e-scale 0 --shuffle"
