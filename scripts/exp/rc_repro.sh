#!/bin/bash
# builds the variants of the rc_packed experiment (on the build box: hipcc cross-compiles) — run once before gpurun
set -e
cd $(dirname $0)/../..
export BNPK_SKIP_ISA_LINT=1
bash scripts/exp/make_variant.sh rc_u4 "-DBNPK_RCP_UNROLL=4"
bash scripts/exp/make_variant.sh rc_u4_nop "-DBNPK_RCP_UNROLL=4 -mllvm -amdgpu-snop-padding=3"
bash scripts/exp/make_variant.sh rc_u4_wz "-DBNPK_RCP_UNROLL=4 -mllvm -amdgpu-waitcnt-forcezero"
bash scripts/exp/make_variant.sh rc_u4_nolds "-DBNPK_RCP_UNROLL=4 -DBNPK_RCP_NO_LDS=1"
bash scripts/exp/make_variant.sh rc_u4_fence "-DBNPK_RCP_UNROLL=4 -DBNPK_RCP_FENCE=1"
bash scripts/exp/make_variant.sh rc_u3 "-DBNPK_RCP_UNROLL=3"
bash scripts/exp/make_variant.sh rc_u4_o2 "-DBNPK_RCP_UNROLL=4 -O2"
