cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python scripts/exp/exp_loop_profile.py 2>&1 | grep -v amdgpu > gpurun_out/loop_profile.txt
cut -c1-160 gpurun_out/loop_profile.txt | head -120
