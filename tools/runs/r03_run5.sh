#!/bin/bash
# Round 3, GPU call 5: the tests the -x stop skipped, fuzz, then the round's profiles: rocprofv3 kernel stats + PMC passes of
# bench.py (profile_bench.sh) and of the 3072 / 4096-bit operations (profile_cmd.sh).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
L=gpurun_out/r03_run5.log; : > $L
timeout 1200 python -m pytest tests/test_gpu_paillier_abi.py tests/test_gpu_transcripts.py tests/test_gpu_bench_multirank.py tests/test_gpu_multi.py tests/test_gpu_modarith.py -m gpu -q 2>&1 | tail -6 >> $L
echo "== fuzz" >> $L
timeout 500 python tools/fuzz_gpu.py 200 2>&1 | grep -v amdgpu.ids | tail -8 >> $L
echo "== profile bench" >> $L
bash tools/profile_bench.sh r03 > /dev/null 2>&1
head -8 gpurun_out/prof_r03/kernel_stats.csv >> $L
cat gpurun_out/prof_r03/pmc_fetch_write.json | cut -c1-600 >> $L
echo "== profile k4096 / k3072" >> $L
bash tools/profile_cmd.sh k4096_r03 python $PWD/tools/keysize_sweep.py --bits 4096 > /dev/null 2>&1
bash tools/profile_cmd.sh k3072_r03 python $PWD/tools/keysize_sweep.py --bits 3072 > /dev/null 2>&1
head -10 gpurun_out/prof_k4096_r03/kernel_stats.csv >> $L
head -10 gpurun_out/prof_k3072_r03/kernel_stats.csv >> $L
cut -c1-260 $L
