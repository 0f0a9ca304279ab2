#!/bin/bash
# round 4, GPU call 16: g-factored PAIR tables (3072 / 4096-bit keys): parity tests and kernel times with and without
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_keysizes.py tests/test_gpu_paillier_abi.py -x -q -m gpu -k "3072 or 4096" > gpurun_out/r04_run16_tests.log 2>&1; tail -5 gpurun_out/r04_run16_tests.log
for g in 1 0; do
  PAI_FB_GFORM=$g python tools/keysize_sweep.py --bits 3072 4096 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('gform $g', d['key_bits'], {k:d[k] for k in ('key_setup_s','encrypt_ms','decrypt_ms','roundtrip_ok')})"
  PAI_FB_GFORM=$g python tools/first_call_keysizes.py 2>/dev/null | tail -4
done
