#!/bin/bash
# Dev probe: rebuilds ONE geometry translation unit with extra flags into lib/alt/lib_<tag>.so (the other objects are
# the in-tree build's).  Usage: bash tools/variant_geo.sh <geo, e.g. 36x4> <tag> "<extra hipcc flags>"
set -e
cd "$(dirname "$0")/.."
C=pailliercryptolib_python_amd/csrc
OUT=pailliercryptolib_python_amd/lib/alt
mkdir -p $OUT
BASE="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -pragma-unroll-threshold=1048576"
OTHERS=$(ls $C/build/*.o | grep -v geo_$1.o)
hipcc $BASE $3 -c $C/geo_$1.hip -o $OUT/geo_$1_$2.o
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/lib_$2.so $OTHERS $OUT/geo_$1_$2.o
rm -f $OUT/geo_$1_$2.o
