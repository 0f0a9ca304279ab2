#!/bin/bash
# round 5, GPU call 24: long differential fuzz on the final tree (every C-ABI operation against CPython integers)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 2000 python tools/fuzz_gpu.py 1800 2>&1 | tail -3 | tee gpurun_out/r05_fuzz24.json
