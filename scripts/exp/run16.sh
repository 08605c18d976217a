cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "radix or sparse or count or kmer or pipeline" 2>&1 | tail -2
for line in 16 8 4; do echo "LINE $line"; BNPK_RP_LINE=$line timeout 300 python scripts/exp/exp_l1.py 50000000 9,10 | grep level; done
BNPK_RP_LINE=8 timeout 600 python scripts/exp/exp_levels.py 3000000000 2>&1 | grep "level" | sed -n '3p;6p'
