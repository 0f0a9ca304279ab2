#!/bin/bash
# round 4, GPU call 23: right-to-left small-batch decryption on wave pairs (k_dec_a_rl): parity + latency A/B
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_paillier_abi.py -x -q -m gpu -k "decrypt_latency" > gpurun_out/r04_run23_tests.log 2>&1; tail -5 gpurun_out/r04_run23_tests.log
for b in 2048 1024 3072; do timeout 300 python tools/lat_rl_probe.py $b 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/lat_rl_probe.jsonl; done
