#!/bin/bash
# round 4, GPU call 18: g-factoring with chunks of 256 entries: parity test + first-call cost
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_paillier_abi.py -x -q -m gpu -k "g_factored or cache or trim" > gpurun_out/r04_run18_tests.log 2>&1; tail -3 gpurun_out/r04_run18_tests.log
python tools/first_call_keysizes.py 2>/dev/null | tail -4
python tools/first_call_probe.py 2>/dev/null | tail -4
