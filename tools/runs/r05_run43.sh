#!/bin/bash
# round 5, GPU call 43: mid-size decryption by default: parity tests, the batch-size sweep at 2048 bits, short fuzz
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_paillier_abi.py tests/test_gpu_baseline_configs.py -m gpu -q -x -k "decrypt or config" 2>&1 | tail -3
timeout 600 python tools/latency_sweep.py 2048 dense 2>&1 | grep bits | tee gpurun_out/r05_sweep43.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print(d['bits'], d['N'], 'dec def', d['dec_def_ms'], 'lat', d['dec_lat_ms'], 'thr', d['dec_thr_ms'])
"
timeout 300 python tools/fuzz_gpu.py 150 2>&1 | tail -2
