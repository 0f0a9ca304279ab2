#!/bin/bash
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 > gpurun_out/r06/gputest_after_refactor.log; cat gpurun_out/r06/gputest_after_refactor.log
timeout 900 python bench.py --steps 3 --warmup 1 > gpurun_out/r06/bench_c.json 2> gpurun_out/r06/bench_c.err; echo "bench rc=$?"
tail -c 1200 gpurun_out/r06/bench_c.json
