# other shapes of the headline step through the C planner (round 6): every line must say parity_fullsize true
cd $GRAFT_REPO_ROOT
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-fed --no-extra"
show() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', d.get('value'), d.get('ms_per_step'), d.get('parity_fullsize'), d.get('planner'))"; }
timeout 300 $B --canonical 2>/dev/null | show canonical
timeout 300 $B --k 17 2>/dev/null | show k17
timeout 300 $B --reads 30000000 --read-len 250 2>/dev/null | show 30Mx250
timeout 300 $B --reads 70000000 --read-len 100 2>/dev/null | show 70Mx100
timeout 300 $B --reads 12000000 --read-len 600 --k 27 2>/dev/null | show 12Mx600_k27
timeout 300 $B --reads 40000000 --read-len 151 --k 25 --mode genome --genome-len 3000000 2>/dev/null | show 40Mx151_k25_genome3M
timeout 300 $B --reads 2000000 2>/dev/null | show 2M_reads
timeout 300 $B --reads 200000 --k 31 2>/dev/null | show 200k_reads
timeout 600 python bench.py --from-file /tmp/bnpk_r06.fq --reads 8000000 --steps 2 --warmup 1 2>/dev/null | tail -1 | cut -c1-600
rm -f /tmp/bnpk_r06.fq
