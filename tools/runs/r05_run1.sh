#!/bin/bash
# round 5, GPU call 1: default bench.py with the new configs block; GPU suite with per-test durations (baseline for the suite shrink)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
( time timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench1.json 2> gpurun_out/r05_bench1.err ) 2> gpurun_out/r05_bench1.time
tail -c 600 gpurun_out/r05_bench1.err; cat gpurun_out/r05_bench1.time
timeout 1500 python -m pytest tests -m gpu -q -x --durations=80 > gpurun_out/r05_gputest1.log 2>&1
tail -5 gpurun_out/r05_gputest1.log
