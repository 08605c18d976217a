cd $GRAFT_REPO_ROOT
timeout 300 python scripts/exp/exp_l1.py 50000000 9,10 | grep level
echo half; timeout 300 python scripts/bin/half/scripts/exp/exp_l1.py 50000000 9,10 | grep level
