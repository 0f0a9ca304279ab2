#!/bin/bash
# Sustained v_mad_u64_u32 rate with rocm-smi power / clock samples beside it.  Through gpurun:
#   bash tools/ubench_sustained.sh [seconds_per_config]   ->  gpurun_out/ubench_valu_sustained.jsonl, gpurun_out/ubench_valu_sustained_power.txt
cd "$(dirname "$0")/.."
SEC=${1:-3}
mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/ubench_sustained tools/ubench_sustained.hip 2>/dev/null
OUT=gpurun_out/ubench_valu_sustained.jsonl; PW=gpurun_out/ubench_valu_sustained_power.txt
: > $OUT; : > $PW
rocm-smi --showmaxpower 2>/dev/null | grep -i max >> $PW
for form in 0 1 2; do
  echo "== form $form" >> $PW
  ( tools/ubench_sustained $SEC $form >> $OUT 2>&1 ) &
  BP=$!
  T0=$(date +%s.%N)
  while kill -0 $BP 2>/dev/null; do
    T=$(python3 -c "import time,sys; print(round(time.time()-float(sys.argv[1]),2))" $T0)
    S=$(rocm-smi --showpower --showclocks 2>/dev/null | grep -iE "sclk|Socket Graphics" | sed 's/.*: //' | tr '\n' ' ')
    echo "$T $S" >> $PW
  done
  wait $BP
done
cat $OUT
awk '/^==/{print} !/^==/{print}' $PW | head -120
