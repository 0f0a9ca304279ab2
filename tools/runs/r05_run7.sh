#!/bin/bash
# round 5, GPU call 7: k_dec_a_pp timing after a change (probe only) + the decrypt parity test
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_paillier_abi.py -m gpu -q -x -k "decrypt_latency" 2>&1 | tail -3
for b in 2048 4096; do timeout 300 python tools/lat_pp_probe.py $b 2>&1 | grep bits | head -2; done | tee gpurun_out/r05_lat_pp7.jsonl
