cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_api.py -x -q -m gpu -k "reverse_complement or revcomp or rewrite" 2>&1 | tail -3
timeout 600 python scripts/exp/exp_movers.py 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['rewrite'])"
