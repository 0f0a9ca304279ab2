#!/bin/bash
# round 4, GPU call 4: k_modmul (staged 36x4) experiments: occupancy, both loads in flight, priority, modulus from LDS
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( bash tools/variant_geo.sh 36x4 load2 "-DPAI_MODMUL_LOAD2=1" ) &
( bash tools/variant_geo.sh 36x4 noprio "-DPAI_MODMUL_PRIO=0" ) &
( bash tools/variant_geo.sh 36x4 load2np "-DPAI_MODMUL_LOAD2=1 -DPAI_MODMUL_PRIO=0" ) &
( bash tools/variant_geo.sh 36x4 nmlds "-DPAI_TILE_NMLDS=true" ) &
( bash tools/variant_geo.sh 36x4 nmlds2 "-DPAI_TILE_NMLDS=true -DPAI_MODMUL_LOAD2=1" ) &
wait
: > gpurun_out/ctadd_ab2.jsonl
PAI_DEBUG_OCC=1 python tools/ctadd_ab.py >> gpurun_out/ctadd_ab2.jsonl 2>gpurun_out/ctadd_ab2.err
for v in load2 noprio load2np nmlds nmlds2; do
  PAI_DEBUG_OCC=1 PAI_NATIVE_LIB=$PWD/pailliercryptolib_python_amd/lib/alt/lib_$v.so python tools/ctadd_ab.py >> gpurun_out/ctadd_ab2.jsonl 2>>gpurun_out/ctadd_ab2.err
done
cat gpurun_out/ctadd_ab2.jsonl; grep PAI_OCC gpurun_out/ctadd_ab2.err | sort | uniq -c
timeout 900 python -m pytest tests/test_gpu_paillier_abi.py -x -q -m gpu -k "cache or async or add" > gpurun_out/r04_run4_tests.log 2>&1; tail -4 gpurun_out/r04_run4_tests.log
