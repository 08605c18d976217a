cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05c9
timeout 120 python scripts/exp/vgpr24/run.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05c9/vgpr24.log
