#!/bin/bash
# round 4, GPU call 1: new tests (config lines, 8-rank dry run, full-size API config test, keygen), bench lines for every config
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_bench_multirank.py tests/test_gpu_baseline_configs.py -x -q -m gpu > gpurun_out/r04_run1_tests.log 2>&1
tail -5 gpurun_out/r04_run1_tests.log
python bench.py > gpurun_out/bench_r04a.json 2> gpurun_out/bench_r04a.err; tail -c 600 gpurun_out/bench_r04a.err
python bench.py --config cfg4 --steps 2 > gpurun_out/bench_r04a_cfg4.json 2> gpurun_out/bench_r04a_cfg4.err; tail -c 600 gpurun_out/bench_r04a_cfg4.err
python bench.py --config cfg5 --steps 2 > gpurun_out/bench_r04a_cfg5.json 2> gpurun_out/bench_r04a_cfg5.err; tail -c 600 gpurun_out/bench_r04a_cfg5.err
for f in gpurun_out/bench_r04a*.json; do python - "$f" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel_ms"], (d.get("cpu_baseline") or {}).get("value"))
rb=d.get("reference_bench")
if rb:
    for k,v in rb["rows"].items(): print(k, {a:(round(b,1) if isinstance(b,float) else b) for a,b in v.items() if a!="note"})
PY
done
