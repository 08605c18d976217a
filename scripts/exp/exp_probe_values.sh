cd $GRAFT_REPO_ROOT
COVS=${1:-1 2 3 4 5 6 8 12 20 30 40 60 100}
for c in $COVS; do
  echo "coverage ${c}x: $(BNPK_FINISH_DEBUG=1 timeout 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-host-fed --no-extra --mode genome --genome-len $((7500000000 / c)) 2>&1 >/dev/null | grep 'finish probe' | sort | uniq -c | head -3)"
done
echo "uniform: $(BNPK_FINISH_DEBUG=1 timeout 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-host-fed --no-extra 2>&1 >/dev/null | grep 'finish probe' | sort | uniq -c | head -3)"
echo "k21: $(BNPK_FINISH_DEBUG=1 timeout 300 python bench.py --k 21 --steps 1 --warmup 0 --no-cpu-baseline --no-host-fed --no-extra 2>&1 >/dev/null | grep 'finish probe' | sort | uniq -c | head -3)"
