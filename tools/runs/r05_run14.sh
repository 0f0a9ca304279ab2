#!/bin/bash
# round 5, GPU call 14: ct * pt of the smallest batches on the four-wave digit-pair pipeline (k_ctmul_pp): parity + probe
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_paillier_abi.py -m gpu -q -x -k "ct_mul or decrypt_latency" 2>&1 | tail -8
for b in 2048 1024; do timeout 300 python tools/lat_mul_probe.py $b 2>&1 | grep bits; done | tee gpurun_out/r05_lat_mul14.jsonl
