cd $GRAFT_REPO_ROOT
(cd scripts/bin/ff_phases && FM_SPARE=1 timeout 300 python scripts/exp/exp_modes_k21.py 31 50000000 2 2>&1 | grep -v amdgpu)
