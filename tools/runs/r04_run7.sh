#!/bin/bash
# round 4, GPU call 7: pair kernels with the normalisation every 18 rows (PAIR_NORM_MAX): key-size sweep + the 3072/4096-bit tests
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python tools/keysize_sweep.py --bits 3072 4096 > gpurun_out/keysize_sweep_norm18.jsonl 2> gpurun_out/keysize_sweep_norm18.err
cat gpurun_out/keysize_sweep_norm18.jsonl | cut -c1-600
timeout 1200 python -m pytest tests/test_gpu_keysizes.py tests/test_gpu_baseline_configs.py -x -q -m gpu -k "3072 or 4096 or keysize or other_key" > gpurun_out/r04_run7_tests.log 2>&1; tail -4 gpurun_out/r04_run7_tests.log
