#!/bin/bash
# round 4, GPU call 28: small-batch ct+ct on the latency geometry: parity + timing
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_paillier_abi.py tests/test_gpu_api.py -x -q -m gpu -k "ct_add or lazy_montgomery or sums or chain or align" > gpurun_out/r04_run28_tests.log 2>&1; tail -5 gpurun_out/r04_run28_tests.log
for b in 2048 1024 4096; do timeout 300 python tools/lat_add_probe.py $b 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/lat_add_probe.jsonl; done
