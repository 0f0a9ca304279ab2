#!/bin/bash
cd "$(dirname "$0")/.."
L=gpurun_out/r03_run8.log; : > $L
echo "== latency sweep 2048, 2x64 + sliding windows" >> $L
timeout 600 python tools/latency_sweep.py 2048 2>&1 | grep -v amdgpu.ids | head -5 >> $L
echo "== latency sweep 2048, 3x64 + sliding windows" >> $L
PAI_LAT_GEO3=1 timeout 600 python tools/latency_sweep.py 2048 2>&1 | grep -v amdgpu.ids | head -3 >> $L
echo "== 1024 / 3072 / 4096" >> $L
for b in 1024 3072 4096; do timeout 600 python tools/latency_sweep.py $b 2>&1 | grep -v amdgpu.ids | head -2 >> $L; done
timeout 1500 python -m pytest tests/test_gpu_paillier_abi.py tests/test_gpu_keysizes.py tests/test_gpu_api.py tests/test_gpu_baseline_configs.py -m gpu -x -q 2>&1 | tail -4 >> $L
cut -c1-250 $L
