#!/bin/bash
# round 5, GPU call 38: quotient digit masked on the scalar side
cd "$(dirname "$0")/../.."
timeout 900 python -m pytest tests/test_gpu_paillier_abi.py -m gpu -q -x -k "ct_mul or decrypt_latency" 2>&1 | tail -2
for b in 2048 3072 4096; do timeout 300 python tools/lat_pp_probe.py $b 2>&1 | grep bits | head -2 | cut -c1-110; done
timeout 300 python tools/lat_mul_probe.py 2048 2>&1 | grep bits | head -2
