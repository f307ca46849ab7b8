cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/final
( timeout 1200 python -m pytest tests -m gpu -x -q -p no:faulthandler 2>&1 | tail -3 ) > gpurun_out/final/pytest_gpu.txt
( KD_GUARD=1 timeout 1800 python -m pytest tests -m gpu -x -q -s -p no:faulthandler 2>&1 | tail -4 ) > gpurun_out/final/pytest_guard1.txt
( KD_GUARD=2 timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -s -p no:faulthandler 2>&1 | tail -4 ) > gpurun_out/final/pytest_guard2.txt
for f in gpurun_out/final/*.txt; do echo "== $f"; cat $f | cut -c1-300; done
