#!/bin/bash
# rocprofv3 capture of the default bench command (run on the GPU box via gpurun); summaries are copied
# from gpurun_out/ into profiles/ by scripts/summarize_profile.py
set -x
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
READS=${1:-50000000}
rocprofv3 --kernel-trace --stats -d $OUT/stats -o bench -- python $REPO/bench.py --reads $READS --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_stats.json 2> $OUT/bench_stats.err
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o bench -- python $REPO/bench.py --reads $READS --steps 1 --warmup 0 --no-cpu-baseline > $OUT/bench_fetch.json 2> $OUT/bench_fetch.err
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o bench -- python $REPO/bench.py --reads $READS --steps 1 --warmup 0 --no-cpu-baseline > $OUT/bench_write.json 2> $OUT/bench_write.err
find $OUT -type f | head -50
du -sh $OUT
