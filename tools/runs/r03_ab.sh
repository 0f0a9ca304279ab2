#!/bin/bash
# Round 3 same-box A/B of the streamed-table kernels: library at HEAD~ (lib_head.so), the new library, and the 28x8
# two-waves-per-SIMD variant; then the GPU tests that exercise what changed.
cd "$(dirname "$0")/.."
ALT=$PWD/pailliercryptolib_python_amd/lib/alt
mkdir -p gpurun_out
for v in head new w2; do
  if [ $v = new ]; then unset PAI_NATIVE_LIB; else export PAI_NATIVE_LIB=$ALT/lib_$v.so; fi
  [ $v != new ] && [ ! -f "$PAI_NATIVE_LIB" ] && continue
  echo "== $v" >> gpurun_out/r03_ab.log
  timeout 600 python tools/keysize_sweep.py --bits 3072 4096 >> gpurun_out/r03_ab.log 2>&1
done
unset PAI_NATIVE_LIB
timeout 1500 python -m pytest tests/test_gpu_paillier_abi.py tests/test_gpu_keysizes.py tests/test_gpu_modarith.py tests/test_gpu_api.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r03_gputest2.log
cat gpurun_out/r03_ab.log | cut -c1-400
cat gpurun_out/r03_gputest2.log
