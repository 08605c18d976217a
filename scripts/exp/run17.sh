cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_api.py tests/test_gpu_parity.py -x -q -m gpu -k "row_values or filter or quality or elementwise or reduc or rowops or row_" 2>&1 | tail -3
timeout 500 python scripts/exp/exp_filter.py 2>&1 | tail -13
