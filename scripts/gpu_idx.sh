cd $GRAFT_REPO_ROOT
timeout 120 python scripts/exp/exp_index.py 2>&1 | grep -v amdgpu.ids | head -22
timeout 300 python -m pytest tests/test_fullsize.py tests/test_api.py -q -m gpu -k "index or Index or debruijn or config5" 2>&1 | tail -3
