#!/bin/bash
# PMC counters + kernel trace of k_modmul (tools/ctadd_only.py): one rocprofv3 --pmc pass per counter group.
export TMPDIR=/tmp
TAG=${1:-ctadd}
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p $OUT
R=$PWD
(cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python $R/tools/ctadd_only.py > $OUT/trace.log 2>&1)
DB=$(find $OUT/trace -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB > $OUT/kernel_stats.csv
i=0
for P in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU" "SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  (cd /tmp && rocprofv3 --pmc $P --output-format csv -d $OUT/p$i -o p -- python $R/tools/ctadd_only.py > $OUT/p$i.log 2>&1)
done
python3 tools/pmc_summary.py $OUT/summary.json $(find $OUT -name "*counter_collection.csv") > /dev/null
find $OUT -name "*.db" -delete
find $OUT -name "*counter_collection.csv" -size +1M -delete
python3 - <<PY
import json
d=json.load(open("$OUT/summary.json"))
for k,v in d.items():
    if k.startswith("k_"): print(k, {c:int(x) for c,x in v.items()})
PY
cat $OUT/kernel_stats.csv | head -12
