cd $GRAFT_REPO_ROOT
R=gpurun_out/r05c7
mkdir -p $R
/opt/rocm/bin/rocminfo 2>/dev/null | grep -i "gfx950\|xnack\|Compute Unit\|Marketing" | head -8
echo HSA_XNACK=$HSA_XNACK
v=rc_u4
cp -r tests scripts/bin/$v/tests; cp -r oracle scripts/bin/$v/oracle; cp scripts/exp/rc_repro4.py scripts/bin/$v/scripts/exp/
(cd scripts/bin/$v && timeout 200 python scripts/exp/rc_repro4.py > $GRAFT_REPO_ROOT/$R/${v}_cumask.log 2>&1); grep -v amdgpu.ids $R/${v}_cumask.log | cut -c1-220
