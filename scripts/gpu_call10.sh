cd $GRAFT_REPO_ROOT
R=gpurun_out/r05c10
mkdir -p $R
for v in rc_u4_floor rc_u4_nofloor; do
cp -r tests scripts/bin/$v/tests; cp -r oracle scripts/bin/$v/oracle; cp scripts/exp/rc_repro3.py scripts/bin/$v/scripts/exp/
(cd scripts/bin/$v && timeout 200 python scripts/exp/rc_repro3.py quick > $GRAFT_REPO_ROOT/$R/${v}.log 2>&1); echo "== $v"; grep -v amdgpu.ids $R/${v}.log | cut -c1-160
done
v=rc_u4_floor
(cd scripts/bin/$v && timeout 200 python scripts/exp/fuzz_parity.py 60 1 > $GRAFT_REPO_ROOT/$R/${v}_fuzz.log 2>&1); tail -3 $R/${v}_fuzz.log
