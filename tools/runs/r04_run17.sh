#!/bin/bash
# round 4, GPU call 17: final tree (g-factored tables): profiles of the three configurations, full suite, smoke, fuzz, bench lines
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
bash tools/profile_bench.sh r04 > gpurun_out/profile_bench_r04.log 2>&1
bash tools/profile_cmd.sh cfg4_r04 python $PWD/bench.py --config cfg4 --steps 1 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/profile_cfg4_r04.log 2>&1
bash tools/profile_cmd.sh cfg5_r04 python $PWD/bench.py --config cfg5 --steps 1 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/profile_cfg5_r04.log 2>&1
head -4 gpurun_out/prof_r04/kernel_stats.csv; head -3 gpurun_out/prof_cfg4_r04/kernel_stats.csv; head -3 gpurun_out/prof_cfg5_r04/kernel_stats.csv
timeout 1800 python -m pytest tests -x -q -m gpu > gpurun_out/gputest_r04_final.log 2>&1; tail -5 gpurun_out/gputest_r04_final.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python tools/fuzz_gpu.py 300 > gpurun_out/fuzz_r04.json 2> gpurun_out/fuzz_r04.err; tail -c 300 gpurun_out/fuzz_r04.json
python bench.py > gpurun_out/bench_r04_final.json 2> gpurun_out/bench_r04_final.err; tail -c 300 gpurun_out/bench_r04_final.err
python bench.py --config cfg4 --steps 2 > gpurun_out/bench_r04_final_cfg4.json 2> gpurun_out/bench_r04_final_cfg4.err
python bench.py --config cfg5 --steps 2 > gpurun_out/bench_r04_final_cfg5.json 2> gpurun_out/bench_r04_final_cfg5.err
for f in gpurun_out/bench_r04_final*.json; do python - "$f" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], round(d["value"]), round(d["ms_per_step"],1), round(d["roofline"]["frac"],3), d["roofline"]["kernel_ms"], (d.get("cpu_baseline") or {}).get("value"))
rb=d.get("reference_bench")
if rb:
    for k,v in rb["rows"].items(): print("  ",k, {a:(round(b,1) if isinstance(b,float) else b) for a,b in v.items() if a!="note"})
    print("  api", d["api_level"]); print("  other", {k:round(v) for k,v in d["other_ops"].items() if "ops_per_s" in k})
PY
done
