#!/bin/bash
# round 4, GPU call 10: 72-limb decrypt kernel (4096-bit keys) variants: 12-row blocks, limb-class symmetric squaring
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( bash tools/variant_dec.sh u72_12 -DPADIC_U72=12 ) &
( bash tools/variant_dec.sh u72_12sym -DPADIC_U72=12 -DPADIC_SQR_SYM_MAX_NL=72 ) &
( bash tools/variant_dec.sh u72_4 -DPADIC_U72=4 ) &
wait
for v in default u72_12 u72_12sym u72_4; do
  L=""; [ $v != default ] && L=$PWD/pailliercryptolib_python_amd/lib/alt/lib_$v.so
  PAI_NATIVE_LIB=$L timeout 600 python bench.py --config cfg5 --batch 65536 --no-extras --no-cpu-baseline --steps 2 > gpurun_out/bench_cfg5_$v.json 2> gpurun_out/bench_cfg5_$v.err
  python - "$v" <<'PY'
import json,sys
try:
    d=json.loads(open(f"gpurun_out/bench_cfg5_{sys.argv[1]}.json").read().strip().splitlines()[-1])
    print(sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"])
except Exception as e: print(sys.argv[1], "FAILED", e, open(f"gpurun_out/bench_cfg5_{sys.argv[1]}.err").read()[-800:])
PY
done
