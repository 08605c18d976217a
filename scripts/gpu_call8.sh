cd $GRAFT_REPO_ROOT
R=gpurun_out/r05c8
mkdir -p $R
for v in rc_u4_v32 rc_u4; do
cp -r tests scripts/bin/$v/tests; cp -r oracle scripts/bin/$v/oracle; cp scripts/exp/rc_repro3.py scripts/exp/rc_repro4.py scripts/bin/$v/scripts/exp/
(cd scripts/bin/$v && timeout 200 python scripts/exp/rc_repro3.py quick > $GRAFT_REPO_ROOT/$R/${v}.log 2>&1); echo "== $v"; grep -v amdgpu.ids $R/${v}.log | cut -c1-200
done
