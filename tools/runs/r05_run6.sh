#!/bin/bash
# round 5, GPU call 6: four-wave digit-pair decryption of the smallest batches (k_dec_a_pp): parity at 1024..4096 bits, latency
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_paillier_abi.py tests/test_gpu_keysizes.py -m gpu -q -x -k "decrypt_latency or key_size_boundaries" > gpurun_out/r05_t6.log 2>&1; tail -15 gpurun_out/r05_t6.log
for b in 2048 1024 3072 4096; do timeout 300 python tools/lat_pp_probe.py $b 2>&1 | grep bits; done | tee gpurun_out/r05_lat_pp.jsonl
