#!/bin/bash
# Builds lib/alt/lib_<tag>.so with extra flags on padic_dec_kernels.hip (A/B timing of decrypt-kernel variants):
#   bash tools/variant_dec.sh sym -DPAI_PADIC_SQR_SYM
set -e
cd "$(dirname "$0")/.."
C=pailliercryptolib_python_amd/csrc
OUT=pailliercryptolib_python_amd/lib/alt
mkdir -p $OUT
TAG=$1; shift
BASE="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -pragma-unroll-threshold=1048576"
hipcc $BASE "$@" -c $C/padic_dec_kernels.hip -o $OUT/dec_$TAG.o
OTHERS=$(ls $C/build/*.o | grep -v padic_dec_kernels)
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/lib_$TAG.so $OTHERS $OUT/dec_$TAG.o
rm -f $OUT/dec_$TAG.o
