cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 120 python scripts/exp/exp_l1.py 50000000 8,9,10 > gpurun_out/l1_full.log 2>&1
timeout 120 python scripts/bin/half/scripts/exp/exp_l1.py 50000000 8,9 > gpurun_out/l1_half.log 2>&1
tail -n 4 gpurun_out/l1_full.log gpurun_out/l1_half.log
