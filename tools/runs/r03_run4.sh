#!/bin/bash
# Round 3, GPU call 4: full GPU suite, bench line, fuzz, corrected MFMA probe with power samples.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
L=gpurun_out/r03_run4.log; : > $L
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 >> $L
timeout 900 python bench.py > gpurun_out/r03_bench2.json 2> gpurun_out/r03_bench2.err
tail -c 400 gpurun_out/r03_bench2.err >> $L
echo "== fuzz" >> $L
timeout 400 python tools/fuzz_gpu.py 240 2>&1 | grep -v amdgpu.ids | tail -5 >> $L
echo "== mfma probe" >> $L
bash tools/mfma_probe_power.sh > /dev/null 2>&1
( tools/mfma_probe ) >> $L 2>&1
cut -c1-300 $L
