set -x
cd $GRAFT_REPO_ROOT
R=gpurun_out/r05c4
mkdir -p $R
v=rc_u4
cp -r tests scripts/bin/$v/tests; cp -r oracle scripts/bin/$v/oracle; cp scripts/exp/rc_repro2.py scripts/bin/$v/scripts/exp/
(cd scripts/bin/$v && timeout 200 python scripts/exp/rc_repro2.py 60 > $GRAFT_REPO_ROOT/$R/${v}_seq.log 2>&1); tail -40 $R/${v}_seq.log
(cd scripts/bin/$v && timeout 200 python scripts/exp/rc_repro2.py 40 alone > $GRAFT_REPO_ROOT/$R/${v}_alone.log 2>&1); tail -12 $R/${v}_alone.log
