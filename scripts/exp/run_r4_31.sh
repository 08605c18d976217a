cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gather_encode or encoding_error" 2>&1 | tail -12
timeout 600 python scripts/exp/exp_api_path.py 2>&1 | tail -1
