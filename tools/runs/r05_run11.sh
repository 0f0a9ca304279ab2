#!/bin/bash
# round 5, GPU call 11: randomised differential run (tools/fuzz_gpu.py) on the round's tree
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 900 python tools/fuzz_gpu.py 420 2>&1 | tail -5 | tee gpurun_out/r05_fuzz.json
