cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/b5.log 2>&1
tail -n 1 gpurun_out/b5.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], json.dumps(d['host_fed']))"
