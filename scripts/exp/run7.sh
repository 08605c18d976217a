cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python scripts/exp/exp_levels.py 6000000000 > gpurun_out/e7.log 2>&1
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-fed > gpurun_out/b7.log 2>&1
tail -n 8 gpurun_out/e7.log; tail -n 1 gpurun_out/b7.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['parity_fullsize'], {k:v['ms_per_step'] for k,v in d['kernels'].items()})"
