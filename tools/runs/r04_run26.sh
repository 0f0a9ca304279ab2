#!/bin/bash
# round 4, GPU call 26: wave-shared small-batch DJN encryption (k_encrypt_tree): parity + latency A/B
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_paillier_abi.py -x -q -m gpu -k "djn_encrypt_latency or small_batch_table" > gpurun_out/r04_run26_tests.log 2>&1; tail -5 gpurun_out/r04_run26_tests.log
for b in 2048 1024 4096; do timeout 300 python tools/lat_enc_probe.py $b 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/lat_enc_probe.jsonl; done
