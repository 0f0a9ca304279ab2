#!/bin/bash
cd "$(dirname "$0")/.."
L=gpurun_out/r03_run9.log; : > $L
timeout 1800 python -m pytest tests -m gpu -q --durations=30 2>&1 | tail -45 >> $L
cut -c1-200 $L
