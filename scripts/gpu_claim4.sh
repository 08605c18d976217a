cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05claim
TIMING_ONLY=1 timeout 600 python scripts/exp/exp_claim.py 6e9 uniform 2>&1 | grep -v amdgpu.ids
timeout 600 python scripts/exp/exp_claim.py 6e9 dup 2>&1 | grep -v amdgpu.ids
timeout 600 python scripts/exp/exp_claim.py 3e8 uniform 2>&1 | grep -v amdgpu.ids
