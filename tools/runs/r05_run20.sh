#!/bin/bash
# round 5, GPU call 20: batch-size sweep over the path switches (default against forced paths), keygen rows
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 900 python tools/latency_sweep.py 2048 dense 2>&1 | grep bits | tee gpurun_out/r05_sweep20.jsonl
timeout 300 python - <<'PY' 2>&1 | tail -4
import time, torch
from pailliercryptolib_python_amd import PaillierKeypair, _native
for bits in (1024, 2048):
    PaillierKeypair.generate_keypair(bits)
    t0 = time.perf_counter()
    for _ in range(10): PaillierKeypair.generate_keypair(bits)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(10): _native.keygen(bits, True)
    t2 = time.perf_counter()
    print(bits, "generate_keypair ms", round((t1 - t0) * 100, 2), "pai_keygen alone ms", round((t2 - t1) * 100, 2))
PY
