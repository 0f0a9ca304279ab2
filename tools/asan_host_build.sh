#!/bin/bash
# ASan + UBSan build of the HOST side of the C ABI (csrc/paillier_capi.hip: key set-up, host big integers, key generation,
# shard plans) into lib/alt/lib_asan.so; device code and the other translation units are the in-tree build's.
#   bash tools/asan_host_build.sh && python tools/asan_host_run.py        (sanitizer pass of the native host code)
set -e
cd "$(dirname "$0")/.."
C=pailliercryptolib_python_amd/csrc
OUT=pailliercryptolib_python_amd/lib/alt
mkdir -p $OUT
SAN="-Xarch_host -fsanitize=address,undefined -Xarch_host -fno-omit-frame-pointer -Xarch_host -g"
hipcc --offload-arch=gfx950 -O1 -std=c++17 -fPIC -mllvm -pragma-unroll-threshold=1048576 $SAN -c $C/paillier_capi.hip -o $OUT/paillier_capi_asan.o
OTHERS=$(ls $C/build/*.o | grep -v paillier_capi.o)
hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -shared-libsan -o $OUT/lib_asan.so $OTHERS $OUT/paillier_capi_asan.o
rm -f $OUT/paillier_capi_asan.o
echo $OUT/lib_asan.so
