#!/bin/bash
# round 5, GPU call 9: k_dec_a_pp variants (profile builds): per-wave cycles
cd "$(dirname "$0")/../.."
for v in base nob; do
echo "== $v"
PAI_NATIVE_LIB=$PWD/pailliercryptolib_python_amd/lib/alt/lib_pp_$v.so timeout 120 python - <<'PY' 2>&1 | grep -E "^PP|ok" | sort | uniq -c | head -8
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
out = priv.decrypt(ct); torch.cuda.synchronize()
print("ok", bool(torch.equal(out, m)))
PY
done
timeout 300 python -m pytest tests/test_gpu_paillier_abi.py -m gpu -q -x -k "decrypt_latency" 2>&1 | tail -3
for b in 2048 4096; do timeout 300 python tools/lat_pp_probe.py $b 2>&1 | grep bits | head -2; done
