cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/idxprof -o idx -- python $GRAFT_REPO_ROOT/scripts/exp/exp_index2.py > /tmp/idx.log 2>&1
f=$(find /tmp/idxprof -name "*kernel_stats.csv" | head -1); echo $f; head -40 $f | cut -c1-200
