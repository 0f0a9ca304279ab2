#!/bin/bash
cd "$(dirname "$0")/.."
L=gpurun_out/r03_run10.log; : > $L
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -5 >> $L
timeout 900 python bench.py > gpurun_out/r03_bench4.json 2> gpurun_out/r03_bench4.err
python -c "import __graft_entry__ as g; g.smoke()" >> $L 2>&1
cut -c1-200 $L
