cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_api.py tests/test_gpu_parity.py tests/test_abi.py tests/test_errors.py -x -q -m gpu 2>&1 | tail -3
timeout 600 python scripts/exp/exp_movers.py 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['rewrite'])"
