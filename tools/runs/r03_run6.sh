#!/bin/bash
# Round 3, GPU call 6: same-box A/B of the lane-group pair geometries (ct*pt and DJN encrypt at 3072 / 4096 bits).
cd "$(dirname "$0")/.."
ALT=$PWD/pailliercryptolib_python_amd/lib/alt
L=gpurun_out/r03_run6.log; : > $L
for v in base g14x8 g28u4 g28lds g28u4lds; do
  if [ $v = base ]; then unset PAI_NATIVE_LIB; else export PAI_NATIVE_LIB=$ALT/lib_$v.so; fi
  echo "== 3072 $v" >> $L
  timeout 300 python tools/keysize_sweep.py --bits 3072 2>&1 | grep key_bits | cut -c1-215 >> $L
done
for v in base g18u9 g18u3 g36x4; do
  if [ $v = base ]; then unset PAI_NATIVE_LIB; else export PAI_NATIVE_LIB=$ALT/lib_$v.so; fi
  echo "== 4096 $v" >> $L
  timeout 300 python tools/keysize_sweep.py --bits 4096 2>&1 | grep key_bits | cut -c1-215 >> $L
done
unset PAI_NATIVE_LIB
echo "== latency sweep 2048 (minus-one ct*pt)" >> $L
timeout 600 python tools/latency_sweep.py 2048 2>&1 | grep -v amdgpu.ids | head -4 >> $L
timeout 600 python tools/latency_sweep.py 4096 2>&1 | grep -v amdgpu.ids | head -2 >> $L
timeout 900 python -m pytest tests/test_gpu_paillier_abi.py tests/test_gpu_keysizes.py tests/test_gpu_api.py -m gpu -x -q -k "ct_mul or mul or keysize or key_size or ctmul or times" 2>&1 | tail -4 >> $L
cat $L
