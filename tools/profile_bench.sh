#!/bin/bash
# Round profile of bench.py on the GPU box: kernel trace + stats, then PMC passes (each alone, as the
# microarchitecture guide prescribes).  Usage (from the repo root, through gpurun): bash tools/profile_bench.sh TAG
set -u
TAG=${1:-final}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
B="python $PWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-configs"
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- $B > $OUT/bench_under_rocprofv3.json 2> $OUT/trace.err)
DB=$(find $OUT/trace -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB > $OUT/kernel_stats.csv
i=0
for P in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  (cd /tmp && timeout 900 rocprofv3 --pmc $P --output-format csv -d $OUT/pmc$i -o pmc -- $B > /dev/null 2> $OUT/pmc$i.err)
done
python tools/pmc_summary.py $OUT/pmc_summary.json $(find $OUT -name "*counter_collection.csv") > $OUT/pmc_fetch_write.json
# keep the artefacts small: drop the raw databases / per-dispatch CSVs
find $OUT -name "*.db" -delete
find $OUT -name "*counter_collection.csv" -size +2M -delete
ls -la $OUT
