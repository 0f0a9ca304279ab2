#!/bin/bash
# Clock / power samples (rocm-smi) while each form of the reduction probe runs for seconds: tools/mfma_probe <iters> <mode>.
cd "$(dirname "$0")/.."
hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/mfma_probe tools/mfma_probe.hip 2>/dev/null
OUT=gpurun_out/mfma_probe_power.txt; : > $OUT
for mode in 1 2; do
  echo "== mode $mode (1 = MFMA form, 2 = VALU form)" >> $OUT
  ( tools/mfma_probe 400000 $mode > gpurun_out/mfma_probe_mode$mode.json 2>/dev/null ) &
  BP=$!
  T0=$(date +%s.%N)
  while kill -0 $BP 2>/dev/null; do
    T=$(echo "$(date +%s.%N) - $T0" | bc)
    S=$(rocm-smi --showpower --showclocks 2>/dev/null | grep -iE "sclk|Socket Graphics" | sed 's/.*: //' | tr '\n' ' ')
    echo "$T $S" >> $OUT
  done
  wait $BP
  cat gpurun_out/mfma_probe_mode$mode.json >> $OUT
done
cat $OUT | cut -c1-200
