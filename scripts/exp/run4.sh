cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_fullsize.py -x -q -m gpu --durations=5 2>&1 | tail -25 > gpurun_out/t_full.log
timeout 600 python bench.py --steps 3 --warmup 1 > gpurun_out/b_uniform2.log 2>&1
cat gpurun_out/t_full.log; tail -n 3 gpurun_out/b_uniform2.log
