#!/bin/bash
# round 4, GPU call 24: k_dec_a_rl with cached tail / scalar wait loop / ring 16; floor of the squaring wave (B without products); ring 8
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( bash tools/variant_geo.sh 3x64 rlnob "-DPAI_RL_DEBUG_NOB=1" ) &
( bash tools/variant_geo.sh 3x64 rlring8 "-DPAI_RL_RING=8" ) &
( bash tools/variant_geo.sh 3x64 rlring4 "-DPAI_RL_RING=4" ) &
wait
python tools/lat_rl_probe2.py 2>/dev/null
PAI_LAT_RL=0 python tools/lat_rl_probe2.py 2>/dev/null
for v in rlnob rlring8 rlring4; do PAI_NATIVE_LIB=$PWD/pailliercryptolib_python_amd/lib/alt/lib_$v.so python tools/lat_rl_probe2.py 2>/dev/null; done
timeout 300 python tools/lat_rl_probe.py 2048 2>&1 | grep -v amdgpu.ids
