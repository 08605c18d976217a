#!/bin/bash
# experiment builds of ONE source with extra flags, linked with the product's other objects into
# bionumpy_amd/csrc/variants/libbnpk_<name>.so (loaded with BNPK_LIB=<path>; never the product library)
#   scripts/exp/build_variant.sh <name> <source.hip> <flags...>
set -e
cd "$(dirname "$0")/../.."
name=$1; src=$2; shift 2
C=bionumpy_amd/csrc
mkdir -p $C/variants
obj=$C/variants/${src%.hip}_$name.o
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-result "$@" -c $C/$src -o $obj
others=$(ls $C/build/*.o | grep -v "/${src%.hip}.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o $C/variants/libbnpk_$name.so $obj $others -ldl
echo built $C/variants/libbnpk_$name.so
