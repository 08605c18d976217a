# the whole GPU test suite on the box: gpurun --timeout 2700 -- 'bash scripts/exp/gpu_suite.sh'
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -12 > gpurun_out/t_all.log
cat gpurun_out/t_all.log
