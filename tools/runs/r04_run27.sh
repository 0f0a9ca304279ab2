#!/bin/bash
# round 4, GPU call 27: small-batch ct*pt right to left on wave pairs (k_modexp_rl): parity + latency A/B
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_paillier_abi.py -x -q -m gpu -k "ct_mul_latency" > gpurun_out/r04_run27_tests.log 2>&1; tail -5 gpurun_out/r04_run27_tests.log
for b in 2048 1024 4096; do timeout 300 python tools/lat_mul_probe.py $b 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/lat_mul_probe.jsonl; done
