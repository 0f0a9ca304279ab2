#!/bin/bash
# round 5, GPU call 8: per-wave cycle / wait report of k_dec_a_pp (profile build of geo_3x64)
cd "$(dirname "$0")/../.."
PAI_NATIVE_LIB=$PWD/pailliercryptolib_python_amd/lib/alt/lib_ppprof.so timeout 300 python - <<'PY' 2>&1 | grep -E "^PP|ok" | head -20
import torch, sys
sys.path.insert(0, '.')
from bench import synthetic_key
from pailliercryptolib_python_amd import engine
key = synthetic_key(2048, 0x1234567)
dev = torch.device('cuda', 0)
pub = engine.PublicKeyHandle(key.n, 2048, key.hs, key.randbits, device=dev)
priv = engine.PrivateKeyHandle(pub, key.p, key.q)
g = torch.Generator(device=dev); g.manual_seed(1)
m = torch.randint(0, 2**31 - 1, (16, pub.n_words), dtype=torch.int64, device=dev, generator=g).to(torch.int32)
m[:, -1] &= 0x0FFFFFFF
ct = pub.encrypt(m, pub.random_r(16, generator=g))
priv.decrypt(ct); torch.cuda.synchronize()
print("ok", bool(torch.equal(priv.decrypt(ct), m)))
torch.cuda.synchronize()
PY
