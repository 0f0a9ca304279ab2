#!/bin/bash
# round 5, GPU call 10: whole GPU suite (durations), driver-style bench, rocprofv3 kernel stats + PMC passes, latency probe
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --durations=15 > gpurun_out/r05_gputest10.log 2>&1; tail -22 gpurun_out/r05_gputest10.log
( time timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench10.json 2> gpurun_out/r05_bench10.err ) 2> gpurun_out/r05_bench10.time; tail -3 gpurun_out/r05_bench10.time
bash tools/profile_bench.sh r05b > gpurun_out/r05_profile10.log 2>&1; tail -3 gpurun_out/r05_profile10.log
for b in 2048 3072; do timeout 300 python tools/lat_pp_probe.py $b 2>&1 | grep bits; done | tee gpurun_out/r05_lat_pp10.jsonl
