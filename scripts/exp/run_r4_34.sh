cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_api.py tests/test_reader_big_batches.py tests/test_fullsize.py -x -q -m gpu 2>&1 | tail -4
timeout 600 python scripts/exp/exp_min.py 2>&1 | tail -1
