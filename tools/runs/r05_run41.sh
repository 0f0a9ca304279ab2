#!/bin/bash
# round 5, GPU call 41: long differential fuzz on the final tree, eight key sizes
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 2300 python tools/fuzz_gpu.py 1800 2>&1 | tail -4 | tee gpurun_out/r05_fuzz41.json
