#!/bin/bash
# round 4, GPU call 29: what a product wave / a reduction wave would cost: the minus-one row block without its a*b or without its q*(M+1) products
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( bash tools/variant_geo.sh 3x64 m1noab "-DPAI_M1_DEBUG_HALF=1" ) &
( bash tools/variant_geo.sh 3x64 m1noqn "-DPAI_M1_DEBUG_HALF=2" ) &
wait
python tools/lat_rl_probe2.py 2>/dev/null
for v in m1noab m1noqn; do PAI_NATIVE_LIB=$PWD/pailliercryptolib_python_amd/lib/alt/lib_$v.so python tools/lat_rl_probe2.py 2>/dev/null; done
