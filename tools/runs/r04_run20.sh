#!/bin/bash
# round 4, GPU call 20: PADIC_REGM again, now with the products through mul_wbuf (no spills in the hot loops)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( bash tools/variant_dec.sh regmw -DPADIC_DEC36_MODE=PADIC_REGM -DPADIC_REGM_MUL_WBUF=1 ) &
( bash tools/variant_dec.sh regmwf -DPADIC_DEC36_MODE=PADIC_REGM -DPADIC_REGM_MUL_WBUF=1 -DPADIC_FENCE_CHUNKS=1 ) &
wait
for v in default regmw regmwf; do
  L=""; [ $v != default ] && L=$PWD/pailliercryptolib_python_amd/lib/alt/lib_$v.so
  PAI_NATIVE_LIB=$L python bench.py --no-extras --no-cpu-baseline --steps 4 > gpurun_out/bench_dec_$v.json 2> gpurun_out/bench_dec_$v.err
  python - "$v" <<'PY'
import json,sys
try:
    d=json.loads(open(f"gpurun_out/bench_dec_{sys.argv[1]}.json").read().strip().splitlines()[-1])
    print(sys.argv[1], round(d["value"]), round(d["ms_per_step"],1), d["roofline"]["kernel_ms"])
except Exception as e: print(sys.argv[1], "FAILED", e, open(f"gpurun_out/bench_dec_{sys.argv[1]}.err").read()[-800:])
PY
done
