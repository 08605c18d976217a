cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_fullsize.py -x -q -k "radix or sparse or count or kmer or pipeline or canonical or genome or positions" 2>&1 | tail -2
timeout 300 python scripts/exp/exp_l1.py 50000000 8,9,10 | grep level
