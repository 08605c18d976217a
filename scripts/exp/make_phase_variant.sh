#!/bin/bash
# usage: make_phase_variant.sh <name> <wave>
set -e
REPO=$(cd $(dirname $0)/../.. && pwd)
$REPO/scripts/exp/make_variant.sh $1 "" > /dev/null
V=$REPO/scripts/bin/$1
python $REPO/scripts/exp/add_phase_marks.py $REPO/bionumpy_amd/csrc/radix.hip $V/bionumpy_amd/csrc/radix.hip
sed -i 's|"state\[0:8\]", state\[:8\].tolist())|"phases", [round(x / 256 / 2.4e6, 2) for x in state[8 + nseg + 8: 8 + nseg + 15].tolist()])|' $V/scripts/exp/exp_finish_var.py
BNPK_HIPCC_FLAGS="-DFN_EXP_PHASES=$2" python $V/bionumpy_amd/csrc/build.py --force > /dev/null
ls -la $V/bionumpy_amd/csrc/libbnpk.so
