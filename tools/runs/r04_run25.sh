#!/bin/bash
# round 4, GPU call 25: k_dec_a_rl: poll interval of the product wave (s_sleep 1 / 4 / 8 / 16 / 32)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for v in 1 4 16 32; do ( bash tools/variant_geo.sh 3x64 rlsleep$v "-DPAI_RL_SLEEP_B=$v" ) & done
wait
python tools/lat_rl_probe2.py 2>/dev/null
for v in 1 4 16 32; do PAI_NATIVE_LIB=$PWD/pailliercryptolib_python_amd/lib/alt/lib_rlsleep$v.so python tools/lat_rl_probe2.py 2>/dev/null; done
