#!/bin/bash
# round 5, GPU call 49: long differential fuzz on the final tree (mid-size paths crossed), eight key sizes
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 2000 python tools/fuzz_gpu.py 1500 2>&1 | tail -4 | tee gpurun_out/r05_fuzz49.json
