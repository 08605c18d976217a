cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_api.py tests/test_gpu_parity.py -x -q -m gpu -k "reverse or complement or rewrite" 2>&1 | tail -2
timeout 300 python scripts/exp/exp_rc.py 2>&1 | grep -v amdgpu | tail -1
