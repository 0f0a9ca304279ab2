#!/bin/bash
# PMC counters of the digit-engine probes (tools/padic_bench): one rocprofv3 --pmc pass per counter group.
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_probe
mkdir -p $OUT
i=0
for P in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU" "SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  (cd /tmp && rocprofv3 --pmc $P --output-format csv -d $OUT/p$i -o p -- $OLDPWD/tools/padic_bench 100 > $OUT/p$i.log 2>&1)
done
python3 tools/pmc_summary.py $OUT/summary.json $(find $OUT -name "*counter_collection.csv") > /dev/null
python3 - <<PY
import json
d=json.load(open("$OUT/summary.json"))
for k,v in d.items(): print(k, {c:int(x) for c,x in v.items()})
PY
