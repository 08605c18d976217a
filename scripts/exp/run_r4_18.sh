cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_finish_modes.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
timeout 600 python scripts/exp/exp_modes_k21.py 31 50000000 2,2,5 2>&1 | grep finish_mode
echo "== nogroup"; (cd scripts/bin/ff_nogroup && timeout 300 python scripts/exp/exp_modes_k21.py 31 50000000 2,2 2>&1 | grep finish_mode)
