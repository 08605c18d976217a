cd $GRAFT_REPO_ROOT
R=gpurun_out/r05c6
mkdir -p $R
for v in rc_u4 rc_u4_nop rc_u4_wz rc_u4_nolds rc_u4_fence rc_u3 rc_u4_o2; do
  cp -r tests scripts/bin/$v/tests; cp -r oracle scripts/bin/$v/oracle; cp scripts/exp/rc_repro3.py scripts/bin/$v/scripts/exp/
  (cd scripts/bin/$v && timeout 200 python scripts/exp/rc_repro3.py quick > $GRAFT_REPO_ROOT/$R/${v}.log 2>&1); echo "== $v"; grep -v amdgpu.ids $R/${v}.log | cut -c1-200
done
