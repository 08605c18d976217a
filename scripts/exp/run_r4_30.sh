cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_reader_big_batches.py tests/test_api.py -x -q -m gpu 2>&1 | tail -12
