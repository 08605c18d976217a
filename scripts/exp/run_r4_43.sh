cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_api.py -x -q -m gpu -k "reverse_complement or revcomp or canonical" 2>&1 | tail -2
timeout 300 python scripts/exp/exp_rcp.py 2>&1 | tail -1
