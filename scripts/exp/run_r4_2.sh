cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for a in "2 5000" "6 5000" "11 5700"; do timeout 300 python scripts/exp/dbg_multi.py $a 2>&1 | grep -v amdgpu.ids; echo ----; done > gpurun_out/dbg_multi.log 2>&1
cat gpurun_out/dbg_multi.log
