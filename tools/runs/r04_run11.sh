#!/bin/bash
# round 4, GPU call 11: why is the two-workgroups-per-CU decrypt kernel (PADIC_REGM) slower?  clock / power samples and SQ counters
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
bash tools/variant_dec.sh regm -DPADIC_DEC36_MODE=PADIC_REGM
L=$PWD/pailliercryptolib_python_amd/lib/alt/lib_regm.so
B="python $PWD/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras"
OUT=$PWD/gpurun_out/regm_power.txt; : > $OUT
( PAI_NATIVE_LIB=$L $B > gpurun_out/regm_bench.json 2>/dev/null ) &
BP=$!
sleep 8
for i in $(seq 1 30); do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -iE "sclk|Socket Graphics" | sed 's/.*: //' | tr '\n' ' ' >> $OUT; echo >> $OUT
  kill -0 $BP 2>/dev/null || break
  sleep 0.2
done
wait $BP
sort $OUT | uniq -c | sort -rn | head -12
i=0
for P in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  ( cd /tmp && PAI_NATIVE_LIB=$L timeout 600 rocprofv3 --pmc $P --output-format csv -d $PWD/../gpurun_out_regm_pmc$i -o pmc -- python $OLDPWD/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> /tmp/regm_pmc$i.err )
done
python tools/pmc_summary.py gpurun_out/pmc_regm.json $(find /tmp/../gpurun_out_regm_pmc* /gpurun_out_regm_pmc* -name "*counter_collection.csv" 2>/dev/null) > /dev/null 2>&1
python - <<'PY'
import json
try:
    d=json.load(open("gpurun_out/pmc_regm.json"))
    for k,v in d.items():
        if k.startswith("k_dec_a"): print(k, json.dumps({a:round(b,1) for a,b in v.items()}))
except Exception as e: print("pmc summary failed", e)
PY
