#!/bin/bash
# Clock / power samples (rocm-smi) while the digit-engine probes run for seconds each: one wave per SIMD vs two.
OUT=$PWD/gpurun_out/power_trace_probe.txt
mkdir -p $(dirname $OUT); : > $OUT
( ./tools/padic_bench 150000 > gpurun_out/padic_bench_150k.jsonl 2>/dev/null ) &
BP=$!
T0=$(date +%s.%N)
while kill -0 $BP 2>/dev/null; do
  T=$(echo "$(date +%s.%N) - $T0" | bc)
  S=$(rocm-smi --showpower --showclocks 2>/dev/null | grep -iE "sclk|Socket Graphics" | sed 's/.*: //' | tr '\n' ' ')
  echo "$T $S" >> $OUT
done
wait $BP
cut -c1-70,100-150 gpurun_out/padic_bench_150k.jsonl
cat $OUT
