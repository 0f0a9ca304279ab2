#!/bin/bash
# round 4, GPU call 9: k_modmul with the two waves of a SIMD started in anti-phase (PAI_MODMUL_STAGGER)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for sgr in 3 6 9 12; do ( bash tools/variant_geo.sh 36x4 st$sgr "-DPAI_MODMUL_STAGGER=$sgr" ) & done
wait
: > gpurun_out/ctadd_ab3.jsonl
python tools/ctadd_ab.py >> gpurun_out/ctadd_ab3.jsonl 2>gpurun_out/ctadd_ab3.err
for v in st3 st6 st9 st12; do
  PAI_NATIVE_LIB=$PWD/pailliercryptolib_python_amd/lib/alt/lib_$v.so python tools/ctadd_ab.py >> gpurun_out/ctadd_ab3.jsonl 2>>gpurun_out/ctadd_ab3.err
done
cat gpurun_out/ctadd_ab3.jsonl; tail -3 gpurun_out/ctadd_ab3.err
