cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
echo base; timeout 600 python scripts/exp/exp_movers.py 2>&1 | tail -1
for v in gr_u2 gr_u4; do echo $v; (cd scripts/bin/$v && timeout 600 python scripts/exp/exp_movers.py 2>&1 | tail -1); done
