cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_api.py -x -q -m gpu -k "row_reduc or mean or quality or filter or ragged" 2>&1 | tail -3
timeout 600 python scripts/exp/exp_movers.py 2>&1 | tail -1 | cut -c1-420
