#!/bin/bash
# round 5, GPU call 5: many-keys test, default bench (configs block + table operating points), rocprofv3 kernel stats + PMC passes
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_paillier_abi.py -m gpu -q -x -k "many_keys or cache_evicts" > gpurun_out/r05_t5.log 2>&1; tail -5 gpurun_out/r05_t5.log
bash tools/profile_bench.sh r05 > gpurun_out/r05_profile.log 2>&1; tail -5 gpurun_out/r05_profile.log
python - <<'PY'
import json
d=json.loads(open('gpurun_out/prof_r05/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['fixed_base_table_points'])
print({k:(round(v['value']), round(v['roofline']['frac'],3)) for k,v in d['configs'].items()})
PY
