#!/bin/bash
# round 5, GPU call 46: mid-size ct * pt by default: parity tests (ABI + API + configs), short fuzz
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_paillier_abi.py tests/test_gpu_api.py tests/test_gpu_baseline_configs.py -m gpu -q -x 2>&1 | tail -3
timeout 400 python tools/fuzz_gpu.py 200 2>&1 | tail -2
