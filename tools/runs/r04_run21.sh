#!/bin/bash
# round 4, GPU call 21: g-factoring at 4 products per entry (was 6): parity tests + first-call cost + headline bench
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_paillier_abi.py -x -q -m gpu -k "g_factored or cache or trim or small_batch" > gpurun_out/r04_run21_tests.log 2>&1; tail -3 gpurun_out/r04_run21_tests.log
python tools/first_call_keysizes.py 2>/dev/null | tail -4
python tools/first_call_probe.py 2>/dev/null | tail -4
PAI_FB_GFORM=0 python tools/first_call_probe.py 2>/dev/null | tail -4
python bench.py --no-extras --no-cpu-baseline --steps 4 2> gpurun_out/bench_g4.err | tee gpurun_out/bench_g4.json | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), d['ms_per_step'], d['roofline']['kernel_ms'])"
