set -x
cd $GRAFT_REPO_ROOT
R=gpurun_out/r05c3
mkdir -p $R
for v in rc_u4 rc_u2; do
  cp -r tests scripts/bin/$v/tests; cp -r oracle scripts/bin/$v/oracle
  (cd scripts/bin/$v && timeout 200 python scripts/exp/fuzz_parity.py 100 1 > $GRAFT_REPO_ROOT/$R/${v}_fuzz.log 2>&1); tail -5 $R/${v}_fuzz.log
done
