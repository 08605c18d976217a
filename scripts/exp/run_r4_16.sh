cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 300 python scripts/exp/exp_mall.py 2>&1 | grep -v amdgpu > gpurun_out/mall.json; cat gpurun_out/mall.json
