cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/t_all.log 2>&1
tail -n 4 gpurun_out/t_all.log
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-fed > gpurun_out/b_fq.log 2>&1
tail -n 1 gpurun_out/b_fq.log | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['parity'])
print({k: v['ms_per_step'] for k, v in d['kernels'].items()})
"
