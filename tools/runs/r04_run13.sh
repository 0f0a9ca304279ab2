#!/bin/bash
# round 4, GPU call 13: ct x pt on the digit engine with the limb-class symmetric squaring (PAI_CTMUL_SQR_SYM)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
bash tools/variant_tu.sh ctsym padic_enc_kernels -DPAI_CTMUL_SQR_SYM=true
python tools/ctops_time.py > gpurun_out/ctops_default.json 2>/dev/null; cat gpurun_out/ctops_default.json
PAI_NATIVE_LIB=$PWD/pailliercryptolib_python_amd/lib/alt/lib_ctsym.so python tools/ctops_time.py > gpurun_out/ctops_ctsym.json 2>/dev/null; cat gpurun_out/ctops_ctsym.json
PAI_NATIVE_LIB=$PWD/pailliercryptolib_python_amd/lib/alt/lib_ctsym.so timeout 600 python -m pytest tests/test_gpu_paillier_abi.py -x -q -m gpu -k "mul" > gpurun_out/r04_run13_tests.log 2>&1; tail -3 gpurun_out/r04_run13_tests.log
