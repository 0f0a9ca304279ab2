#!/bin/bash
# Samples GPU clock / power (rocm-smi) while bench.py runs: shows whether the multiplier-dense kernels run at the
# nominal 2.4 GHz or are power-throttled.  Usage (through gpurun): bash tools/power_trace.sh
OUT=$PWD/gpurun_out/power_trace.txt
mkdir -p $(dirname $OUT)
rocm-smi --showmaxpower --showpower --showclocks 2>/dev/null | grep -iE "sclk|power|Max" > $OUT
( python bench.py --no-cpu-baseline --steps 6 --warmup 1 > gpurun_out/power_bench.json 2>/dev/null ) &
BP=$!
sleep 6
for i in $(seq 1 40); do
  echo "--- t=$i" >> $OUT
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -iE "sclk|mclk|power|Temperature \(Sensor (edge|junction)" >> $OUT
  kill -0 $BP 2>/dev/null || break
  sleep 0.25
done
wait $BP
tail -c 300 gpurun_out/power_bench.json | cut -c1-200
grep -iE "sclk|Average Graphics|Current Socket" $OUT | sort | uniq -c | sort -rn | head -20
