cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05claim
timeout 300 python scripts/exp/exp_claim.py 3e8 uniform 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05claim/u3e8.log
timeout 300 python scripts/exp/exp_claim.py 3e8 dup 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05claim/d3e8.log
timeout 300 python scripts/exp/exp_claim.py 7e6 uniform 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05claim/u7e6.log
