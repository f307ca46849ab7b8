#!/bin/bash
# Round 3: the one-wavefront-per-workgroup k_prep: reads per step 4 / 2 / 1, occupancy 4 / 5 / 6 / 8, reads per lane 64 / 32
cd "${GRAFT_REPO_ROOT:-/root/repo}"
bash scripts/gpu_variants.sh "-:N4:--e2e-scale 0" "N2:N2:--e2e-scale 0" "N2o6:N2o6:--e2e-scale 0" "N1o8:N1o8:--e2e-scale 0"
KD_PREP_PER=32 bash scripts/gpu_variants.sh "N2:N2per32:--e2e-scale 0" "-:N4per32:--e2e-scale 0"
KD_PREP_PER=16 bash scripts/gpu_variants.sh "N2:N2per16:--e2e-scale 0"
bash scripts/gpu_variants.sh "N2:N2_C4:--e2e-scale 0 --config C4" "N2:N2_C2:--e2e-scale 0 --config C2" "N2:N2_C5:--e2e-scale 0 --config C5"
