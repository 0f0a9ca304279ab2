#!/bin/bash
# round 5, GPU call 37: mont_mul_m1 in the row form (every minus-one latency kernel): parity + probes
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_paillier_abi.py -m gpu -q -x -k "latency or ct_mul or small or lat or keysize or other_key" 2>&1 | tail -3
timeout 300 python tools/lat_enc_probe.py 2048 2>&1 | grep bits | tee gpurun_out/r05_lat_enc37.jsonl
timeout 300 python tools/lat_mul_probe.py 2048 2>&1 | grep bits | head -4
timeout 300 python tools/lat_pp_probe.py 2048 2>&1 | grep bits | head -2
timeout 300 python tools/lat_add_probe.py 2>&1 | grep -i "bits\|N" | head -6
