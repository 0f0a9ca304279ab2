#!/bin/bash
# Round 3, GPU call 3: minus-one latency decrypt, lane-group digit-pair ct*pt, MFMA probe.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
L=gpurun_out/r03_run3.log; : > $L
echo "== mfma probe" >> $L
( hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/mfma_probe tools/mfma_probe.hip 2>/dev/null; timeout 300 tools/mfma_probe ) >> $L 2>&1
echo "== latency sweep 2048" >> $L
timeout 600 python tools/latency_sweep.py 2048 2>&1 | grep -v amdgpu.ids | head -6 >> $L
echo "== latency sweep 4096" >> $L
timeout 600 python tools/latency_sweep.py 4096 2>&1 | grep -v amdgpu.ids | head -4 >> $L
echo "== keysize sweep" >> $L
timeout 900 python tools/keysize_sweep.py --bits 3072 4096 2>&1 | grep -v amdgpu.ids >> $L
echo "== keysize sweep, pair ct*pt off" >> $L
PAI_DISABLE_PAIR_CTMUL=1 timeout 900 python tools/keysize_sweep.py --bits 3072 4096 2>&1 | grep -v amdgpu.ids | cut -c1-220 >> $L
echo "== tests" >> $L
timeout 1500 python -m pytest tests/test_gpu_keysizes.py tests/test_gpu_paillier_abi.py tests/test_gpu_baseline_configs.py -m gpu -x -q 2>&1 | tail -15 >> $L
cut -c1-420 $L
