#!/bin/bash
# round 4, GPU call 12: final-tree validation: full GPU suite, smoke, fuzz, bench lines of every configuration
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -x -q -m gpu > gpurun_out/gputest_r04_final.log 2>&1; tail -5 gpurun_out/gputest_r04_final.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python tools/fuzz_gpu.py 400 > gpurun_out/fuzz_r04.json 2> gpurun_out/fuzz_r04.err; tail -c 400 gpurun_out/fuzz_r04.json
python bench.py > gpurun_out/bench_r04_final.json 2> gpurun_out/bench_r04_final.err; tail -c 300 gpurun_out/bench_r04_final.err
python bench.py --config cfg4 --steps 2 > gpurun_out/bench_r04_final_cfg4.json 2> gpurun_out/bench_r04_final_cfg4.err
python bench.py --config cfg5 --steps 2 > gpurun_out/bench_r04_final_cfg5.json 2> gpurun_out/bench_r04_final_cfg5.err
for f in gpurun_out/bench_r04_final*.json; do python - "$f" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], round(d["value"]), round(d["ms_per_step"],1), round(d["roofline"]["frac"],3), d["roofline"]["traffic_source"]["source"], (d.get("cpu_baseline") or {}).get("value"))
rb=d.get("reference_bench")
if rb:
    for k,v in rb["rows"].items(): print("  ",k, {a:(round(b,1) if isinstance(b,float) else b) for a,b in v.items() if a!="note"})
    print("  small", d["small_batch"])
PY
done
