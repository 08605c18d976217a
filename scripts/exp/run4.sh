cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -x -q -m gpu --durations=8 2>&1 | tail -25 > gpurun_out/t_all2.log
cat gpurun_out/t_all2.log
