cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_finish_modes.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/t_finish.log
timeout 600 python scripts/exp/exp_modes_k21.py 21 50000000 0,1,5 > gpurun_out/k21.log 2>&1
timeout 600 python scripts/exp/exp_modes_k21.py 31 50000000 2,5,0 > gpurun_out/k31.log 2>&1
cat gpurun_out/t_finish.log gpurun_out/k21.log gpurun_out/k31.log
