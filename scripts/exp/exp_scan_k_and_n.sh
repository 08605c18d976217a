# the headline step at other k and other batch sizes: ms, Gbases/s, which finishing path, the planner's notes
cd $GRAFT_REPO_ROOT
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-fed --no-extra"
show() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1])
    k=d.get('kernels') or {}
    p=d.get('planner') or {}
    print('$1', d.get('value'), d.get('ms_per_step'), d.get('parity_fullsize'), 'path', p.get('path'), 'levels', p.get('levels'), 'trips', p.get('round_trips'), 'bag', p.get('bag'), 'pre', p.get('precounted'),
          {n: round(v['ms_per_step'],1) for n,v in k.items() if n.startswith('finish.') or 'scatter' in n or n.startswith('count') or n.startswith('radix_hist')})
except Exception as e: print('$1', 'failed', e)"; }
for k in ${KS:-14 15 16 17 19 23 25 27 29}; do timeout 300 $B --k $k 2>/dev/null | show "k=$k"; done
for r in ${RS:-1 5 10 20 26 27 30 40}; do timeout 300 $B --reads ${r}000000 2>/dev/null | show "reads=${r}M"; done
