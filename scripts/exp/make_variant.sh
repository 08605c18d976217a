#!/bin/bash
# usage: make_variant.sh <name> "<extra hipcc flags>"   -> scripts/bin/<name>/ = a copy of the package built with the flags
set -e
REPO=$(cd $(dirname $0)/../.. && pwd)
V=$REPO/scripts/bin/$1
rm -rf $V; mkdir -p $V/scripts
cp -r $REPO/bionumpy_amd $V/bionumpy_amd
cp -r $REPO/include $V/include
cp -r $REPO/scripts/exp $V/scripts/exp
cp $REPO/scripts/microbench.py $V/scripts/
rm -rf $V/bionumpy_amd/csrc/build $V/bionumpy_amd/csrc/libbnpk.so $V/bionumpy_amd/csrc/*.sha256
BNPK_HIPCC_FLAGS="$2" python $V/bionumpy_amd/csrc/build.py --force > /dev/null
ls -la $V/bionumpy_amd/csrc/libbnpk.so
