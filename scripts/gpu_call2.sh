set -x
cd $GRAFT_REPO_ROOT
R=gpurun_out/r05c2
mkdir -p $R
timeout 900 python -m pytest tests -m gpu -q > $R/pytest.log 2>&1; tail -8 $R/pytest.log
# the rc_packed experiment under the conditions that found it: the fuzzers, in the variant with the unrolled loop
for v in rc_u4; do
  cp -r tests scripts/bin/$v/tests
  (cd scripts/bin/$v && timeout 200 python scripts/exp/fuzz_parity.py 90 1 > $GRAFT_REPO_ROOT/$R/${v}_fuzz.log 2>&1); tail -5 $R/${v}_fuzz.log
done
