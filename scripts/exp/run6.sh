cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu --durations=6 2>&1 | tail -22 > gpurun_out/t6.log
timeout 600 python bench.py --virtual-ranks 8 --reads 8000000 --steps 1 --warmup 1 > gpurun_out/b6_virtual.log 2>&1
cat gpurun_out/t6.log; tail -n 2 gpurun_out/b6_virtual.log | cut -c1-1500
