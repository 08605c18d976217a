cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for v in fm_abl1 fm_abl2 fm_abl5 fm_lb1 fm_lb4; do echo "== $v"; (cd scripts/bin/$v && timeout 300 python scripts/exp/exp_modes_k21.py 31 50000000 5 2>&1 | grep finish_mode); done > gpurun_out/fm_variants.log 2>&1
cat gpurun_out/fm_variants.log
