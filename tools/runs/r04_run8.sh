#!/bin/bash
# round 4, GPU call 8: the round's rocprofv3 profiles (kernel stats + PMC passes) for the headline configuration and for
# --config cfg4 / cfg5 at their bench sizes, the ct-op kernels, then the bench lines of every configuration
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
bash tools/profile_bench.sh r04 > gpurun_out/profile_bench_r04.log 2>&1
bash tools/profile_cmd.sh cfg4_r04 python $PWD/bench.py --config cfg4 --steps 1 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/profile_cfg4_r04.log 2>&1
bash tools/profile_cmd.sh cfg5_r04 python $PWD/bench.py --config cfg5 --steps 1 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/profile_cfg5_r04.log 2>&1
bash tools/profile_cmd.sh ctops_r04 python $PWD/tools/ctops_time.py > gpurun_out/profile_ctops_r04.log 2>&1
head -8 gpurun_out/prof_r04/kernel_stats.csv; head -6 gpurun_out/prof_cfg4_r04/kernel_stats.csv; head -6 gpurun_out/prof_cfg5_r04/kernel_stats.csv; head -8 gpurun_out/prof_ctops_r04/kernel_stats.csv
du -sh gpurun_out/prof_*
