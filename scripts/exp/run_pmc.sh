#!/bin/bash
# usage: run_pmc.sh <outdir-under-gpurun_out> <n> <b1> <b2>
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $REPO/scripts/exp/exp_radix.py $2 $3 $4 2 > $OUT/timing.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/pmc1 -o x -- python $REPO/scripts/exp/exp_radix.py $2 $3 $4 1 > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc2 -o x -- python $REPO/scripts/exp/exp_radix.py $2 $3 $4 1 > $OUT/pmc2.log 2>&1
find $OUT -name "*.csv" | head; cat $OUT/timing.log
