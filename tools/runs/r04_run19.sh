#!/bin/bash
# round 4, GPU call 19: last check of the committed tree: full GPU suite + smoke + default bench line
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -x -q -m gpu > gpurun_out/gputest_r04_last.log 2>&1; tail -4 gpurun_out/gputest_r04_last.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > gpurun_out/bench_r04_last.json 2> gpurun_out/bench_r04_last.err; tail -c 200 gpurun_out/bench_r04_last.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/bench_r04_last.json").read().strip().splitlines()[-1])
print(round(d["value"]), round(d["ms_per_step"],1), round(d["roofline"]["frac"],3), d["roofline"]["kernel_ms"])
PY
