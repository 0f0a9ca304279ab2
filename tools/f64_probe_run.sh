#!/bin/bash
# FP64-limb probe beside the integer digit-pair squaring (same box, same call): bit-exactness, time per squaring, board power.
cd "$(dirname "$0")/.."
O=gpurun_out/r06_f64; mkdir -p $O
timeout 120 ./tools/f64_probe 64 $O/f64_dump.json > $O/f64_probe.jsonl 2>&1
python tools/f64_probe_check.py $O/f64_dump.json > $O/f64_check.jsonl 2>&1; echo "check rc=$?" >> $O/f64_check.jsonl
timeout 120 ./tools/padic_bench 200 > $O/padic_bench.jsonl 2>&1
# power: ~4 s of each kernel with rocm-smi sampled beside it
for which in f64_1wave f64_2waves int; do
  : > $O/power_$which.txt
  case $which in
    f64_1wave) ( timeout 60 ./tools/f64_probe 200000 - 1 > $O/long_$which.jsonl 2>&1 ) & ;;
    f64_2waves) ( timeout 60 ./tools/f64_probe 100000 - 2 > $O/long_$which.jsonl 2>&1 ) & ;;
    int) ( timeout 60 ./tools/padic_bench 100000 > $O/long_$which.jsonl 2>&1 ) & ;;
  esac
  BP=$!
  while kill -0 $BP 2>/dev/null; do
    rocm-smi --showpower --showclocks 2>/dev/null | grep -iE "sclk|Socket Graphics" | sed 's/.*: //' | tr '\n' ' ' >> $O/power_$which.txt
    echo >> $O/power_$which.txt
  done
  wait $BP
done
cat $O/f64_probe.jsonl $O/f64_check.jsonl; cut -c1-200 $O/padic_bench.jsonl | head -3
for w in f64_1wave f64_2waves int; do echo "== $w"; cat $O/long_$w.jsonl | cut -c1-260 | head -2; sort $O/power_$w.txt | uniq -c | sort -rn | head -3; done
