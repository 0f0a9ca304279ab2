#!/bin/bash
# Builds lib/alt/lib_<tag>.so with extra flags on ONE translation unit (A/B timing of kernel variants on one box):
#   bash tools/variant_tu.sh nopf pair_kernels "-DPAIR_PREFETCH=0"
# then  PAI_NATIVE_LIB=$PWD/pailliercryptolib_python_amd/lib/alt/lib_nopf.so python tools/keysize_sweep.py --bits 4096
set -e
cd "$(dirname "$0")/.."
C=pailliercryptolib_python_amd/csrc
OUT=pailliercryptolib_python_amd/lib/alt
mkdir -p $OUT
TAG=$1; TU=$2; shift; shift
BASE="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -pragma-unroll-threshold=1048576"
hipcc $BASE "$@" -c $C/$TU.hip -o $OUT/${TU}_$TAG.o
OTHERS=$(ls $C/build/*.o | grep -v "/$TU.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/lib_$TAG.so $OTHERS $OUT/${TU}_$TAG.o
rm -f $OUT/${TU}_$TAG.o
