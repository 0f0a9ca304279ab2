#!/bin/bash
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_paillier_abi.py tests/test_gpu_path_edges.py tests/test_gpu_fuzz.py tests/test_gpu_api.py -m gpu -q -x 2>&1 | tail -4
timeout 900 python bench.py --steps 3 --warmup 1 > gpurun_out/r06/bench_e.json 2> gpurun_out/r06/bench_e.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06/bench_e.json').read().strip().splitlines()[-1])
o=d['other_ops']; print({k:round(v) for k,v in o.items() if isinstance(v,(int,float))}, o['ct_add_kernel'], o['ct_add_roofline'])
print(d['configs_summary'])
PY
