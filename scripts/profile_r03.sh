#!/bin/bash
# usage: profile_r03.sh [reads] [uniform|genome] [suffix of gpurun_out/prof]
# rocprofv3 captures of the default bench command (run on the GPU box via gpurun); summaries are copied from
# gpurun_out/ into profiles/ by scripts/summarize_profile.py.  PMC passes are separate runs with --kernel-trace only.
set -x
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof${3:-}
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
READS=${1:-50000000}
MODE=${2:-uniform}
COMMON="--reads $READS --mode $MODE --no-cpu-baseline --no-host-fed"
rocprofv3 --kernel-trace --stats -d $OUT/stats -o bench -- python $REPO/bench.py $COMMON --steps 2 --warmup 1 > $OUT/bench_stats.json 2> $OUT/bench_stats.err
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o bench -- python $REPO/bench.py $COMMON --steps 1 --warmup 0 > $OUT/bench_fetch.json 2> $OUT/bench_fetch.err
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o bench -- python $REPO/bench.py $COMMON --steps 1 --warmup 0 > $OUT/bench_write.json 2> $OUT/bench_write.err
# the host-fed leg: kernel + memory-copy trace (hipMemcpyAsync rows next to the kernels that overlap them); uniform mode only
if [ "$MODE" = "uniform" ]; then
rocprofv3 --kernel-trace --memory-copy-trace --stats -d $OUT/hostfed -o bench --output-format csv -- python $REPO/bench.py --reads $READS --no-cpu-baseline --steps 1 --warmup 0 > $OUT/bench_hostfed.json 2> $OUT/bench_hostfed.err
fi
find $OUT -type f | head -60
du -sh $OUT
