cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
(time timeout 1500 python bench.py --steps 5 --warmup 2) > gpurun_out/bench_full.log 2>&1
tail -5 gpurun_out/bench_full.log | cut -c1-3000
