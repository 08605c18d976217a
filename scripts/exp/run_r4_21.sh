cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_finish_modes.py tests/test_gpu_parity.py -x -q -m gpu -k "finish or heavy or sparse" 2>&1 | tail -3
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-fed --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], {k:v['ms_per_step'] for k,v in d['kernels'].items()})"
