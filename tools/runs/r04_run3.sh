#!/bin/bash
# round 4, GPU call 3: k_modmul_w A/B (18x8 split at 3 / 4 waves per SIMD, 36x4 NMLDS, old staged kernel), correctness of the add paths
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( bash tools/variant_geo.sh 36x4 w4 "-DPAI_MODMUL_W_WAVES=4" ) &
( bash tools/variant_geo.sh 36x4 w2 "-DPAI_MODMUL_W_WAVES=2" ) &
( bash tools/variant_geo.sh 36x4 nosplit "-DPAI_MODMUL_W_SPLIT8=false" ) &
( bash tools/variant_geo.sh 36x4 old "-DPAI_MODMUL_W=false" ) &
wait
: > gpurun_out/ctadd_ab.jsonl
python tools/ctadd_ab.py >> gpurun_out/ctadd_ab.jsonl 2>gpurun_out/ctadd_ab.err
for v in w4 w2 nosplit old; do
  PAI_NATIVE_LIB=$PWD/pailliercryptolib_python_amd/lib/alt/lib_$v.so python tools/ctadd_ab.py >> gpurun_out/ctadd_ab.jsonl 2>>gpurun_out/ctadd_ab.err
done
cat gpurun_out/ctadd_ab.jsonl; tail -3 gpurun_out/ctadd_ab.err
timeout 900 python -m pytest tests/test_gpu_paillier_abi.py tests/test_gpu_modarith.py -x -q -m gpu > gpurun_out/r04_run3_tests.log 2>&1; tail -4 gpurun_out/r04_run3_tests.log
