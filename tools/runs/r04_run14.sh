#!/bin/bash
# round 4, GPU call 14: quotient-digit broadcast of the 8-lane geometries through ds_swizzle_b32 (PAI_BCAST8_SWIZZLE): pair kernels,
# 36x8 / 28x8 lane-group kernels; and the new codec test
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
C=pailliercryptolib_python_amd/csrc; OUT=pailliercryptolib_python_amd/lib/alt; mkdir -p $OUT
BASE="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -pragma-unroll-threshold=1048576 -DPAI_BCAST8_SWIZZLE=1"
for tu in pair_kernels geo_36x8 geo_28x8; do ( hipcc $BASE -c $C/$tu.hip -o $OUT/${tu}_sw.o ) & done; wait
OTHERS=$(ls $C/build/*.o | grep -v "/pair_kernels.o" | grep -v "/geo_36x8.o" | grep -v "/geo_28x8.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/lib_sw8.so $OTHERS $OUT/pair_kernels_sw.o $OUT/geo_36x8_sw.o $OUT/geo_28x8_sw.o
python tools/keysize_sweep.py --bits 3072 4096 > gpurun_out/keysize_default.jsonl 2>/dev/null
PAI_NATIVE_LIB=$PWD/$OUT/lib_sw8.so python tools/keysize_sweep.py --bits 3072 4096 > gpurun_out/keysize_sw8.jsonl 2>/dev/null
python - <<'PY'
import json
for f in ("default","sw8"):
    for l in open(f"gpurun_out/keysize_{f}.jsonl"):
        d=json.loads(l); print(f, d["key_bits"], {k:d[k] for k in ("encrypt_ms","decrypt_ms","ct_add_ms","ct_mul53_ms","ct_invert_ms","roundtrip_ok")})
PY
PAI_NATIVE_LIB=$PWD/$OUT/lib_sw8.so timeout 1200 python -m pytest tests/test_gpu_keysizes.py tests/test_gpu_paillier_abi.py -x -q -m gpu -k "3072 or 4096" > gpurun_out/r04_run14_tests.log 2>&1; tail -3 gpurun_out/r04_run14_tests.log
timeout 600 python -m pytest tests/test_gpu_codec.py -x -q -m gpu > gpurun_out/r04_run14_codec.log 2>&1; tail -3 gpurun_out/r04_run14_codec.log
