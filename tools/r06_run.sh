#!/bin/bash
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 > gpurun_out/r06/gputest_final2.log; cat gpurun_out/r06/gputest_final2.log
timeout 900 python bench.py > gpurun_out/r06/bench_f.json 2> gpurun_out/r06/bench_f.err; echo "bench rc=$?"
tail -c 900 gpurun_out/r06/bench_f.json
