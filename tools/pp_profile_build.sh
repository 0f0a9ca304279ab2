#!/bin/bash
# Dev: builds lib/alt/lib_ppprof.so = the in-tree objects with geo_3x64 recompiled under -DPP_PROFILE (kernels_declat.hpp:
# per-wave cycle / wait report of k_dec_a_pp); run with PAI_NATIVE_LIB=$PWD/pailliercryptolib_python_amd/lib/alt/lib_ppprof.so
set -e
cd "$(dirname "$0")/.."
C=pailliercryptolib_python_amd/csrc; A=pailliercryptolib_python_amd/lib/alt; mkdir -p $A
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -pragma-unroll-threshold=1048576 -DPP_PROFILE "$@" -c $C/geo_3x64.hip -o /tmp/geo_3x64_prof.o
OBJS=$(ls $C/build/*.o | grep -v geo_3x64.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $A/lib_ppprof.so $OBJS /tmp/geo_3x64_prof.o
ls -la $A/lib_ppprof.so
