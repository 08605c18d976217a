cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05claim
timeout 600 python scripts/exp/exp_claim.py 6e9 uniform 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05claim/u6e9.log
timeout 600 python scripts/exp/exp_claim.py 6e9 dup 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05claim/d6e9.log
