#!/bin/bash
# round 5, GPU call 39: closing run on the final tree — whole GPU suite (durations), driver-style bench, rocprofv3 kernel stats +
# PMC passes, small-batch probes, fuzz
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --durations=15 > gpurun_out/r05_gputest39.log 2>&1; tail -22 gpurun_out/r05_gputest39.log
( time timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench39.json 2> gpurun_out/r05_bench39.err ) 2> gpurun_out/r05_bench39.time; tail -3 gpurun_out/r05_bench39.time
bash tools/profile_bench.sh r05d > gpurun_out/r05_profile39.log 2>&1; tail -3 gpurun_out/r05_profile39.log
for b in 2048 3072 4096; do timeout 300 python tools/lat_pp_probe.py $b 2>&1 | grep bits; done | tee gpurun_out/r05_lat_pp39.jsonl | cut -c1-120
timeout 300 python tools/lat_enc_probe.py 2048 2>&1 | grep bits | tee gpurun_out/r05_lat_enc39.jsonl
timeout 700 python tools/fuzz_gpu.py 420 2>&1 | tail -3 | tee gpurun_out/r05_fuzz39.json
