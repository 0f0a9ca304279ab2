#!/bin/bash
# round 5, GPU call 21: batch-size sweep over the path switches at the other key sizes
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
for b in 1024 3072 4096; do timeout 900 python tools/latency_sweep.py $b dense 2>&1 | grep bits; done | tee gpurun_out/r05_sweep21.jsonl
