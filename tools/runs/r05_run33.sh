#!/bin/bash
# round 5, GPU call 33: roles rotated between the workgroups of a CU: decrypt of 16 .. 1024 ciphertexts on the pipeline against the other small-batch kernels
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_paillier_abi.py -m gpu -q -x -k "ct_mul or decrypt_latency" 2>&1 | tail -3
timeout 600 python tools/lat_pp_probe.py 2048 wide 2>&1 | grep bits | tee gpurun_out/r05_lat_pp33.jsonl
