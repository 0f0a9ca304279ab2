#!/bin/bash
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 > gpurun_out/r06/gputest_final.log; cat gpurun_out/r06/gputest_final.log
timeout 300 python tools/ctadd_div_time.py > gpurun_out/r06/ctadd_div_time.jsonl 2>&1; tail -1 gpurun_out/r06/ctadd_div_time.jsonl
timeout 900 python bench.py --steps 3 --warmup 1 > gpurun_out/r06/bench_d.json 2> gpurun_out/r06/bench_d.err; echo "bench rc=$?"
tail -c 1500 gpurun_out/r06/bench_d.json
