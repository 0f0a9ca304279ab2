#!/bin/bash
# round 4, GPU call 5: decrypt stage A with the digit pair only in LDS (PADIC_REGM: two workgroups per CU) against the default
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
bash tools/variant_dec.sh regm -DPADIC_DEC36_MODE=PADIC_REGM
for v in default regm; do
  L=""; [ $v != default ] && L=$PWD/pailliercryptolib_python_amd/lib/alt/lib_$v.so
  PAI_NATIVE_LIB=$L python bench.py --no-extras --no-cpu-baseline --steps 5 > gpurun_out/bench_dec_$v.json 2> gpurun_out/bench_dec_$v.err
  python - "$v" <<'PY'
import json,sys
d=json.loads(open(f"gpurun_out/bench_dec_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print(sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel_ms"])
PY
done
PAI_NATIVE_LIB=$PWD/pailliercryptolib_python_amd/lib/alt/lib_regm.so timeout 900 python -m pytest tests/test_gpu_paillier_abi.py tests/test_gpu_keysizes.py tests/test_gpu_baseline_configs.py -x -q -m gpu -k "decrypt or config1 or roundtrip or keysize or 1024" > gpurun_out/r04_run5_tests.log 2>&1; tail -4 gpurun_out/r04_run5_tests.log
( bash tools/power_trace.sh > /dev/null 2>&1 ); grep -iE "sclk|Socket" gpurun_out/power_trace.txt | sed 's/.*: //' | paste - - | sort | uniq -c | sort -rn | head -8
