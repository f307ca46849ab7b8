#!/bin/bash
# Round 3: status words one cache line apart (KDS_STRIDE 16 / 64): round 2's k_prep (exp/libkd_headS16.so) and the one-wavefront k_prep
cd "${GRAFT_REPO_ROOT:-/root/repo}"
bash scripts/gpu_variants.sh "headS16:headS16:--e2e-scale 0" "-:N4S16:--e2e-scale 0" "N2S16:N2S16:--e2e-scale 0" "N2S64:N2S64:--e2e-scale 0"
KD_PREP_PER=32 bash scripts/gpu_variants.sh "N2S16:N2S16per32:--e2e-scale 0" "N2S64:N2S64per32:--e2e-scale 0"
KD_PREP_PER=16 bash scripts/gpu_variants.sh "N2S64:N2S64per16:--e2e-scale 0"
