cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05claim
export TIMING_ONLY=1
echo "== product"; timeout 600 python scripts/exp/exp_claim.py 6e9 uniform 2>&1 | grep -v amdgpu.ids
for v in abl32 abl64; do cp scripts/exp/exp_claim.py scripts/bin/$v/scripts/exp/; echo "== $v"; (cd scripts/bin/$v && timeout 600 python scripts/exp/exp_claim.py 6e9 uniform 2>&1 | grep -v amdgpu.ids); done
