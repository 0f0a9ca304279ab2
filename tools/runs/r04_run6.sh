#!/bin/bash
# round 4, GPU call 6: k_dec_a_padic<36> variants (time-boxed pass): row-block size 4, window widths 5 / 7
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( bash tools/variant_dec.sh u4 -DPADIC_U36=4 ) &
( bash tools/variant_dec.sh w5 -DPAI_PADIC_SLIDE_BITS=5 ) &
wait
for v in default u4; do
  L=""; [ $v != default ] && L=$PWD/pailliercryptolib_python_amd/lib/alt/lib_$v.so
  PAI_NATIVE_LIB=$L python bench.py --no-extras --no-cpu-baseline --steps 5 > gpurun_out/bench_dec_$v.json 2> gpurun_out/bench_dec_$v.err
  python - "$v" <<'PY'
import json,sys
try:
    d=json.loads(open(f"gpurun_out/bench_dec_{sys.argv[1]}.json").read().strip().splitlines()[-1])
    print(sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel_ms"])
except Exception as e: print(sys.argv[1], "FAILED", e, open(f"gpurun_out/bench_dec_{sys.argv[1]}.err").read()[-600:])
PY
done
