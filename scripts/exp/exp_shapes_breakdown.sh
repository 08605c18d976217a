# kernel breakdown of the shapes that lie off the headline's line (canonical k-mers, batches around 6.8e9 keys)
cd $GRAFT_REPO_ROOT
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-fed --no-extra"
show() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d.get('kernels') or {}
print('$1', d.get('value'), d.get('ms_per_step'), d.get('parity_fullsize'), d.get('planner'))
print('   ', {n: round(v['ms_per_step'],1) for n,v in k.items()})"; }
for shape in ${1:-canonical 12Mx600 57M}; do
  case $shape in
    canonical) timeout 300 $B --canonical 2>/dev/null | show canonical;;
    12Mx600) timeout 300 $B --reads 12000000 --read-len 600 --k 27 2>/dev/null | show 12Mx600_k27;;
    *M) timeout 300 $B --reads ${shape%M}000000 2>/dev/null | show ${shape}_reads;;
    *Mg) timeout 300 $B --reads ${shape%Mg}000000 --mode genome 2>/dev/null | show ${shape}_reads_genome60x;;
  esac
done
