#!/bin/bash
# Round 3: k_prep variants (sums-only CIGAR scan; exact scan out of line / inline; reads per step 4 / 2 / 1; occupancy 4 / 6 / 8) on C3, C4 at 5 steps
cd "${GRAFT_REPO_ROOT:-/root/repo}"
bash scripts/gpu_variants.sh "-:prepA:--e2e-scale 0" "prepB:prepB:--e2e-scale 0" "prepU2:prepU2:--e2e-scale 0" "prepU2o6:prepU2o6:--e2e-scale 0" "prepU1o8:prepU1o8:--e2e-scale 0" "-:prepA_C4:--e2e-scale 0 --config C4" "-:prepA_C2:--e2e-scale 0 --config C2"
