# which finishing path the low-coverage shapes should take: the multiplicity kernel (5), the workgroup table (3) or the whole
# cascade (4), against what mode 0 picks, on reads of a genome at several coverages   usage: exp_finish_rules.sh "4 5 8" "0 3 5"
cd $GRAFT_REPO_ROOT
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-fed --no-extra"
show() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d.get('kernels') or {}
print('$1', d.get('ms_per_step'), d.get('parity_fullsize'), {n: round(v['ms_per_step'],1) for n,v in k.items() if n.startswith('finish')})"; }
COVS=${1:-3 6}
MODES=${2:-0 3 4 5}
for c in $COVS; do
  for m in $MODES; do
    BNPK_FINISH_MODE=$m timeout 300 $B --mode genome --genome-len $((7500000000 / c)) 2>/dev/null | show "coverage ${c}x mode$m"
  done
done
