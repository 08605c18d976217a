cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_errors.py -x -q -m gpu 2>&1 | tail -4
timeout 600 python scripts/exp/exp_api_path.py 2>&1 | tail -1
