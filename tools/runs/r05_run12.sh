#!/bin/bash
# round 5, GPU call 12: latency kernels after the digit prefetch in mont_mul_m1: parity of every small-batch path, timings
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_paillier_abi.py tests/test_gpu_keysizes.py tests/test_gpu_api.py -m gpu -q -x -k "latency or key_size_boundaries or small or lat or reference or tile or g_factored or standard" 2>&1 | tail -3
timeout 300 python tools/lat_pp_probe.py 2048 2>&1 | grep bits | head -2
timeout 600 python bench.py --steps 2 --warmup 1 --no-configs --no-cpu-baseline > gpurun_out/r05_bench12.json 2>gpurun_out/r05_bench12.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_bench12.json').read().strip().splitlines()[-1])
print(d['value'], d['small_batch'])
for k,v in d['reference_bench']['rows'].items(): print(k, round(v['gpu_api_us'],1))
PY
