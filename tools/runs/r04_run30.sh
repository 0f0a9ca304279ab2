#!/bin/bash
# round 4, GPU call 30: wave-shared small-batch encryption on a minus-one context (converted table): parity + latency
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_paillier_abi.py -x -q -m gpu -k "djn_encrypt_latency or small_batch_table or trim or cache" > gpurun_out/r04_run30_tests.log 2>&1; tail -5 gpurun_out/r04_run30_tests.log
for b in 2048 1024 3072; do timeout 300 python tools/lat_enc_probe.py $b 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/lat_enc_m1_probe.jsonl; done
