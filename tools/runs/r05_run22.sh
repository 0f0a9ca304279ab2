#!/bin/bash
# round 5, GPU call 22: per-key-size path switches: default against forced paths again, latency-path parity tests, keygen
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_paillier_abi.py tests/test_gpu_api.py -m gpu -q -x -k "latency or keypair or keygen or generate" 2>&1 | tail -3
for b in 1024 2048 3072 4096; do timeout 900 python tools/latency_sweep.py $b dense 2>&1 | grep bits; done | tee gpurun_out/r05_sweep22.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l)
    if d['N'] < 2048: continue
    print(d['bits'], d['N'], 'dec', d['dec_def_ms'], min(d['dec_lat_ms'], d['dec_thr_ms']), '| enc', d['enc_def_ms'], min(d['enc_lat_ms'], d['enc_thr_ms']), '| mul', d['mul_def_ms'], min(d['mul_lat_ms'], d['mul_thr_ms']))
"
timeout 300 python - <<'PY' 2>&1 | tail -3
import time, torch
from pailliercryptolib_python_amd import PaillierKeypair, _native
for bits in (1024, 2048):
    PaillierKeypair.generate_keypair(bits)
    t0 = time.perf_counter()
    for _ in range(10): PaillierKeypair.generate_keypair(bits)
    t1 = time.perf_counter()
    for _ in range(10): _native.keygen(bits, True)
    t2 = time.perf_counter()
    print(bits, "generate_keypair ms", round((t1 - t0) * 100, 2), "pai_keygen alone ms", round((t2 - t1) * 100, 2))
PY
