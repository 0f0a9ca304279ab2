#!/bin/bash
# Round profile of an arbitrary command on the GPU box: rocprofv3 kernel trace + stats, then PMC passes (each counter
# group in its own run, no tracing domains besides the kernel trace — as MI355X_MICROARCH.md prescribes).
#   bash tools/profile_cmd.sh TAG python tools/keysize_sweep.py --bits 3072
# Writes gpurun_out/prof_TAG/{kernel_stats.csv,pmc_summary.json,cmd.out}.
set -u
TAG=$1
shift
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
( cd /tmp && timeout 1200 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- "$@" > $OUT/cmd.out 2> $OUT/trace.err )
DB=$(find $OUT/trace -name "*.db" | head -1)
[ -n "$DB" ] && python $R/tools/rocpd_summary.py $DB > $OUT/kernel_stats.csv
i=0
for P in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  ( cd /tmp && timeout 1200 rocprofv3 --pmc $P --output-format csv -d $OUT/pmc$i -o pmc -- "$@" > /dev/null 2> $OUT/pmc$i.err )
done
python $R/tools/pmc_summary.py $OUT/pmc_summary.json $(find $OUT -name "*counter_collection.csv") > /dev/null
find $OUT -name "*.db" -delete
find $OUT -name "*counter_collection.csv" -delete
find $OUT -name "*agent_info.csv" -delete
head -12 $OUT/kernel_stats.csv
