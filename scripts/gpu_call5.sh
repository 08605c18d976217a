set -x
cd $GRAFT_REPO_ROOT
R=gpurun_out/r05c5
mkdir -p $R
v=rc_u4
cp -r tests scripts/bin/$v/tests; cp -r oracle scripts/bin/$v/oracle; cp scripts/exp/rc_repro3.py scripts/bin/$v/scripts/exp/
(cd scripts/bin/$v && timeout 200 python scripts/exp/rc_repro3.py > $GRAFT_REPO_ROOT/$R/${v}_scrub.log 2>&1); tail -50 $R/${v}_scrub.log
