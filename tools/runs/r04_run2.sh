#!/bin/bash
# round 4, GPU call 2: LDS-direct load probe, k_modmul phase probe, full GPU suite, bench line
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 tools/lds_direct_probe.hip -o /tmp/lds_probe 2>/dev/null && /tmp/lds_probe > gpurun_out/lds_direct_probe.txt 2>&1
cat gpurun_out/lds_direct_probe.txt
bash tools/phase_probe.sh > gpurun_out/phase_build.log 2>&1
PAI_NATIVE_LIB=$PWD/pailliercryptolib_python_amd/lib/alt/lib_phase.so python tools/ctops_time.py > gpurun_out/phase_probe.txt 2>&1
grep -E "PHASES|ct_add" gpurun_out/phase_probe.txt | sort | uniq -c | sort -rn | head -12
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r04_run2_tests.log 2>&1
tail -6 gpurun_out/r04_run2_tests.log
python bench.py --no-cpu-baseline > gpurun_out/bench_r04b.json 2> gpurun_out/bench_r04b.err; tail -c 400 gpurun_out/bench_r04b.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/bench_r04b.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"])
for k,v in d["reference_bench"]["rows"].items(): print(k, {a:(round(b,1) if isinstance(b,float) else b) for a,b in v.items() if a!="note"})
print({k:v for k,v in d["other_ops"].items() if "ops_per_s" in k})
print(d["small_batch"])
PY
