#!/bin/bash
# usage: profile_r04.sh <suffix of gpurun_out/prof> <bench.py arguments ...>
# rocprofv3 captures of one bench command (run on the GPU box via gpurun): kernel trace + stats, then FETCH_SIZE and
# WRITE_SIZE in passes of their own (--kernel-trace only next to --pmc).  scripts/summarize_profile.py turns the result into
# profiles/<name>_kernel_stats.txt and profiles/<name>_pmc.json.
set -x
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof$1
shift
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
COMMON="--no-cpu-baseline --no-host-fed --no-extra $*"
rocprofv3 --kernel-trace --stats -d $OUT/stats -o bench -- python $REPO/bench.py $COMMON --steps 2 --warmup 1 > $OUT/bench_stats.json 2> $OUT/bench_stats.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o bench -- python $REPO/bench.py $COMMON --steps 1 --warmup 0 > $OUT/bench_fetch.json 2> $OUT/bench_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o bench -- python $REPO/bench.py $COMMON --steps 1 --warmup 0 > $OUT/bench_write.json 2> $OUT/bench_write.err
du -sh $OUT
