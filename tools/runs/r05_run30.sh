#!/bin/bash
# round 5, GPU call 30: rows a multiple of four again (the tail of the row loop stays for generality): parity, probes
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_paillier_abi.py -m gpu -q -x -k "ct_mul or decrypt_latency" 2>&1 | tail -4
for b in 2048 3072 4096; do timeout 300 python tools/lat_pp_probe.py $b 2>&1 | grep bits | head -4; done | tee gpurun_out/r05_lat_pp30.jsonl
timeout 300 python tools/lat_mul_probe.py 2048 2>&1 | grep bits | head -3
PAI_NATIVE_LIB=$PWD/pailliercryptolib_python_amd/lib/alt/lib_ppprof.so timeout 300 python tools/lat_pp_probe.py 2048 2>&1 | grep "^PP" | tail -6
