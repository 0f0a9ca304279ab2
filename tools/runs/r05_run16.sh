#!/bin/bash
# round 5, GPU call 16: short-quotient way out of the minus-one contexts (m1_reduce_to_true_modulus), borrow look-ahead: latency-path parity + probes
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_paillier_abi.py -m gpu -q -x -k "latency or ct_mul or small or lat" 2>&1 | tail -4
for b in 2048; do timeout 300 python tools/lat_mul_probe.py $b 2>&1 | grep bits | head -5; done | tee gpurun_out/r05_lat_mul16.jsonl
timeout 300 python tools/lat_pp_probe.py 2048 2>&1 | grep bits | head -3
PAI_NATIVE_LIB=$PWD/pailliercryptolib_python_amd/lib/alt/lib_ppprof.so timeout 300 python - <<'PY' 2>&1 | grep -v "^$" | tail -12
import os, sys, torch
sys.path.insert(0, os.getcwd())
from bench import synthetic_key
from pailliercryptolib_python_amd import engine
dev = torch.device('cuda', 0)
key = synthetic_key(2048, 0x1234567)
pub = engine.PublicKeyHandle(key.n, 2048, key.hs, key.randbits, device=dev)
g = torch.Generator(device=dev); g.manual_seed(1)
N = 1
m = torch.randint(0, 2**31 - 1, (N, pub.n_words), dtype=torch.int64, device=dev, generator=g).to(torch.int32)
m[:, -1] &= 0x0FFFFFFF
ct = pub.encrypt(m, pub.random_r(N, generator=g))
e = torch.randint(-2**31, 2**31 - 1, (N, 2), dtype=torch.int64, device=dev, generator=g).to(torch.int32)
e[:, 1] &= (1 << 21) - 1
e[:, 1] |= 1 << 20
pub.ct_mul(ct, e, 53); torch.cuda.synchronize()
print("---- second call", flush=True)
pub.ct_mul(ct, e, 53); torch.cuda.synchronize()
PY
timeout 300 python tools/lat_enc_probe.py 2048 2>&1 | grep bits | head -4
timeout 300 python tools/api_small.py 2>&1 | tail -12
