#!/bin/bash
# usage: run_trace.sh <outdir-under-gpurun_out> <reads>
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o x -- python $REPO/scripts/microbench.py $2 2 > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/pmc1 -o x -- python $REPO/scripts/microbench.py $2 1 > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc2 -o x -- python $REPO/scripts/microbench.py $2 1 > $OUT/pmc2.log 2>&1
ls $OUT/trace; head -30 $OUT/trace/x_kernel_stats.csv | cut -c1-200
