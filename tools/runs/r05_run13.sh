#!/bin/bash
# round 5, GPU call 13: k_dec_a_pp beyond one workgroup per CU (N = 160 .. 512) against the wave-pair / window kernels
cd "$(dirname "$0")/../.."
timeout 300 python - <<'PY' 2>&1 | grep -v amdgpu
import json, os, sys, time
sys.path.insert(0, '.')
import torch
from bench import synthetic_key
from pailliercryptolib_python_amd import engine
dev = torch.device('cuda', 0)
key = synthetic_key(2048, 0x1234567)
pub = engine.PublicKeyHandle(key.n, 2048, key.hs, key.randbits, device=dev)
priv = engine.PrivateKeyHandle(pub, key.p, key.q)
def tm(f, reps=5):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
g = torch.Generator(device=dev); g.manual_seed(1)
for N in (128, 160, 192, 256, 320, 512):
    m = torch.randint(0, 2**31 - 1, (N, pub.n_words), dtype=torch.int64, device=dev, generator=g).to(torch.int32)
    m[:, -1] &= 0x0FFFFFFF
    ct = pub.encrypt(m, pub.random_r(N, generator=g))
    row = {"N": N}
    for name, tune in (("pp", "lat_pp=100000"), ("default_no_pp", "lat_pp=0")):
        os.environ["PAI_TUNE"] = tune
        ok = bool(torch.equal(priv.decrypt(ct), m))
        row[name] = {"ok": ok, "ms": round(tm(lambda: priv.decrypt(ct)), 3)}
    print(json.dumps(row), flush=True)
PY
