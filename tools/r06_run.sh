#!/bin/bash
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_api.py tests/test_gpu_transcripts.py -m gpu -q -x 2>&1 | tail -5
timeout 300 python -m pytest tests/test_gpu_paillier_abi.py -m gpu -q -x -k "host_stage" 2>&1 | tail -3
timeout 900 python bench.py --steps 2 --warmup 1 --no-configs --no-cpu-baseline > gpurun_out/r06/bench_b.json 2> gpurun_out/r06/bench_b.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06/bench_b.json').read().strip().splitlines()[-1])
print(json.dumps(d.get('reference_bench_summary_us')))
print(json.dumps(d.get('small_batch')))
PY
