#!/bin/bash
# Round 3, GPU call 7: full GPU suite and the bench line on the final tree.
cd "$(dirname "$0")/.."
L=gpurun_out/r03_run7.log; : > $L
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -6 >> $L
timeout 900 python bench.py > gpurun_out/r03_bench3.json 2> gpurun_out/r03_bench3.err
tail -c 300 gpurun_out/r03_bench3.err >> $L
python -c "import __graft_entry__ as g; g.smoke()" >> $L 2>&1
cut -c1-300 $L
