#!/bin/bash
# Dev probe: builds variants of the library for the 36x4 geometry (2048-bit keys) into lib/alt/:
#   lib_phase.so   k_modmul prints cycle counts per phase of its tile loop (-DPAI_PHASE_TIMING)
#   lib_nmlds.so   tile-I/O kernels with the modulus slice re-read from LDS (-DPAI_TILE_NMLDS=true)
# Run on the GPU box:  PAI_NATIVE_LIB=$PWD/pailliercryptolib_python_amd/lib/alt/lib_X.so python tools/ctops_time.py
set -e
cd "$(dirname "$0")/.."
C=pailliercryptolib_python_amd/csrc
OUT=pailliercryptolib_python_amd/lib/alt
mkdir -p $OUT
BASE="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -pragma-unroll-threshold=1048576"
OTHERS=$(ls $C/build/*.o | grep -v geo_36x4)
variant() {  # tag, extra flags
  hipcc $BASE $2 -c $C/geo_36x4.hip -o $OUT/geo_36x4_$1.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/lib_$1.so $OTHERS $OUT/geo_36x4_$1.o
  rm -f $OUT/geo_36x4_$1.o
}
variant phase "-DPAI_PHASE_TIMING" &
variant nmlds "-DPAI_TILE_NMLDS=true" &
variant nostore "-DPAI_PHASE_TIMING -DPAI_PROBE_NOSTORE" &
wait
