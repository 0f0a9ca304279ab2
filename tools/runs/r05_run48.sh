#!/bin/bash
# round 5, GPU call 48: closing run on the final tree (with the mid-size paths) — whole GPU suite (durations), driver-style bench, rocprofv3 kernel stats +
# PMC passes, small-batch probes, fuzz
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --durations=15 > gpurun_out/r05_gputest48.log 2>&1; tail -22 gpurun_out/r05_gputest48.log
( time timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench48.json 2> gpurun_out/r05_bench48.err ) 2> gpurun_out/r05_bench48.time; tail -3 gpurun_out/r05_bench48.time
bash tools/profile_bench.sh r05e > gpurun_out/r05_profile48.log 2>&1; tail -3 gpurun_out/r05_profile48.log
for b in 2048 3072 4096; do timeout 300 python tools/lat_pp_probe.py $b 2>&1 | grep bits; done | tee gpurun_out/r05_lat_pp48.jsonl | cut -c1-120
timeout 300 python tools/lat_enc_probe.py 2048 2>&1 | grep bits | tee gpurun_out/r05_lat_enc48.jsonl
timeout 700 python tools/fuzz_gpu.py 420 2>&1 | tail -3 | tee gpurun_out/r05_fuzz48.json
timeout 600 python tools/latency_sweep.py 2048 dense 2>&1 | grep bits | tee gpurun_out/r05_sweep48.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print(d['N'], 'dec', d['dec_def_ms'], 'enc', d['enc_def_ms'], 'mul', d['mul_def_ms'])
"
