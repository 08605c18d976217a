#!/bin/bash
# builds the variants of the rc_packed experiment (on the build box: hipcc cross-compiles) — run once before gpurun
set -e
cd $(dirname $0)/../..
BNPK_SKIP_ISA_LINT=1 bash scripts/exp/make_variant.sh rc_u4 "-DBNPK_RCP_UNROLL=4"
BNPK_SKIP_ISA_LINT=1 bash scripts/exp/make_variant.sh rc_u4_pad "-DBNPK_RCP_UNROLL=4 -DBNPK_RCP_LDS_PAD=57344"
BNPK_SKIP_ISA_LINT=1 bash scripts/exp/make_variant.sh rc_u4_o1 "-DBNPK_RCP_UNROLL=4 -O1"
BNPK_SKIP_ISA_LINT=1 bash scripts/exp/make_variant.sh rc_u2 "-DBNPK_RCP_UNROLL=2"
