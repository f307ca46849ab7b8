cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/final
for i in 1 2; do ( timeout 500 python scripts/exp/flake_hunt.py 150 aged 2>&1 | grep -a "aged by\|flake hunt\|iter \|channels hit" | cut -c1-500 ) ; done | tee gpurun_out/final/flake_hunt_aged.txt
