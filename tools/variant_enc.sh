#!/bin/bash
# Builds lib/alt/lib_<tag>.so with extra flags on padic_enc_kernels.hip AND padic_dec_kernels.hip (A/B timing of
# digit-pair kernel variants):   bash tools/variant_enc.sh fused "-DPAI_FUSED_ENCRYPT=true" "-DPADIC_FUSED_72=true"
set -e
cd "$(dirname "$0")/.."
C=pailliercryptolib_python_amd/csrc
OUT=pailliercryptolib_python_amd/lib/alt
mkdir -p $OUT
TAG=$1; ENCF=$2; DECF=$3
BASE="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -pragma-unroll-threshold=1048576"
hipcc $BASE $ENCF -c $C/padic_enc_kernels.hip -o $OUT/enc_$TAG.o &
hipcc $BASE $DECF -c $C/padic_dec_kernels.hip -o $OUT/dec_$TAG.o &
wait
OTHERS=$(ls $C/build/*.o | grep -v padic_dec_kernels | grep -v padic_enc_kernels)
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/lib_$TAG.so $OTHERS $OUT/enc_$TAG.o $OUT/dec_$TAG.o
rm -f $OUT/enc_$TAG.o $OUT/dec_$TAG.o
