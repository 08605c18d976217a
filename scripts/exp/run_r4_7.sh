cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_finish_modes.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/t_finish.log
timeout 600 python scripts/exp/exp_modes_k21.py 21 50000000 0,5 > gpurun_out/k21.log 2>&1
timeout 600 python scripts/exp/exp_modes_k21.py 31 50000000 2,5 > gpurun_out/k31.log 2>&1
for v in fm_abl1 fm_cw fm_d2; do echo "== $v"; (cd scripts/bin/$v && FM_SPARE=1 timeout 300 python scripts/exp/exp_modes_k21.py 31 50000000 5 2>&1 | grep -v amdgpu); done > gpurun_out/fm_variants.log 2>&1
cat gpurun_out/t_finish.log gpurun_out/k21.log gpurun_out/k31.log gpurun_out/fm_variants.log
