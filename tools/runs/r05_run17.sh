#!/bin/bash
# round 5, GPU call 17: per-wave cycle stamps of k_ctmul_pp (profile build)
cd "$(dirname "$0")/../.."
PAI_NATIVE_LIB=$PWD/pailliercryptolib_python_amd/lib/alt/lib_ppprof.so timeout 300 python - <<'PY' 2>&1 | grep -v "^$" | tail -12
import os, sys, torch, time
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
t0 = time.perf_counter()
pub.ct_mul(ct, e, 53); torch.cuda.synchronize()
print("wall ms", (time.perf_counter() - t0) * 1e3)
PY
