#!/bin/bash
# round 5, GPU call 36: range of the ct * pt pipeline; latency-path parity tests on the new decrypt range
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 300 python tools/lat_mul_probe.py 2048 2>&1 | grep bits | tee gpurun_out/r05_lat_mul36.jsonl
timeout 900 python -m pytest tests/test_gpu_paillier_abi.py tests/test_gpu_api.py -m gpu -q -x 2>&1 | tail -3
