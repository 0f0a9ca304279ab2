#!/bin/bash
# round 4, GPU call 15: g-factored fixed-base tables (4 NL^2 table products): parity tests, encrypt kernel time, first-call time
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_paillier_abi.py tests/test_gpu_keysizes.py tests/test_gpu_transcripts.py -x -q -m gpu -k "encrypt or obfusc or table or trim or 1024 or transcript or cache" > gpurun_out/r04_run15_tests.log 2>&1; tail -5 gpurun_out/r04_run15_tests.log
for g in 1 0; do
  PAI_FB_GFORM=$g python bench.py --no-extras --no-cpu-baseline --steps 3 > gpurun_out/bench_gform$g.json 2> gpurun_out/bench_gform$g.err
  python - "$g" <<'PY'
import json,sys
try:
    d=json.loads(open(f"gpurun_out/bench_gform{sys.argv[1]}.json").read().strip().splitlines()[-1])
    print("gform", sys.argv[1], round(d["value"]), round(d["ms_per_step"],1), d["roofline"]["kernel_ms"])
except Exception as e: print("gform", sys.argv[1], "FAILED", e, open(f"gpurun_out/bench_gform{sys.argv[1]}.err").read()[-1500:])
PY
  PAI_FB_GFORM=$g python tools/first_call_probe.py 2>/dev/null | tail -3
done
