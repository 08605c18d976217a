cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_finish_modes.py -x -q -m gpu 2>&1 | tail -3
timeout 600 python scripts/microbench.py 50000000 3 2>&1 | grep -v amdgpu | tail -22
