cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python scripts/exp/exp_finish2.py 6000000000 3 noparity 2 > gpurun_out/e9_6e9.log 2>&1
timeout 600 python scripts/exp/exp_finish2.py 3000000000 3 noparity 2 > gpurun_out/e9_3e9.log 2>&1
tail -n 2 gpurun_out/e9_6e9.log; tail -n 2 gpurun_out/e9_3e9.log
