cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_reader_big_batches.py -x -q -m gpu -k "random_files" 2>&1 | tail -8
