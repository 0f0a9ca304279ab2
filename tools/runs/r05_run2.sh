#!/bin/bash
# round 5, GPU call 2: pai_ct_addn parity + timing; per-result inversion outcome tests
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_paillier_abi.py -m gpu -q -x -k "addn or sticky" > gpurun_out/r05_t2.log 2>&1; tail -15 gpurun_out/r05_t2.log
timeout 600 python -m pytest tests/test_gpu_api.py -m gpu -q -x > gpurun_out/r05_t2b.log 2>&1; tail -15 gpurun_out/r05_t2b.log
for k in 2 4 8 16; do timeout 300 python tools/addn_time.py --k $k; done 2>&1 | tee gpurun_out/r05_addn_time.jsonl
PAI_DEBUG_OCC=1 timeout 300 python tools/addn_time.py --k 8 --batch 65536 2>&1 | grep PAI_OCC | sort | uniq
