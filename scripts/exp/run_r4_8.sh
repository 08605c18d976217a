cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for v in fm_abl17 fm_abl21 fm_abl5; do echo "== $v"; (cd scripts/bin/$v && FM_SPARE=1 timeout 300 python scripts/exp/exp_modes_k21.py 31 50000000 2,5 2>&1 | grep -v amdgpu); done > gpurun_out/fm_variants.log 2>&1
cat gpurun_out/fm_variants.log
