"""API-level a - b and ct^(2^52) on 2^20 ciphertexts (2048-bit key), with pai_ct_pow2 on the digit engine (default) and on
the lane-group kernel (PAI_POW2_DIGIT_MIN=huge):  python tools/probe_sub.py"""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
from bench import synthetic_key
from pailliercryptolib_python_amd import PaillierPublicKey
from pailliercryptolib_python_amd.bindings import ipclPublicKey
key = synthetic_key(2048, 0x1234567)
pk = PaillierPublicKey(ipclPublicKey(key.n, 2048, True, hs=key.hs, randbits=key.randbits))
B = 1 << 20
rng = np.random.default_rng(1)
x, y = rng.uniform(-1000, 1000, B), rng.uniform(-1000, 1000, B)
ex, ey = pk.encrypt(x), pk.encrypt(y)
h = pk.pubkey.handle
def tm(f, reps=3):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return round((time.perf_counter() - t0) / reps * 1e3, 2)
out = {}
d52 = torch.full((1,), 52, dtype=torch.int32, device=h.device)
d12 = torch.full((1,), 12, dtype=torch.int32, device=h.device)
for name, env in (("digit", None), ("lane_group", str(1 << 40))):
    if env is None: os.environ.pop("PAI_POW2_DIGIT_MIN", None)
    else: os.environ["PAI_POW2_DIGIT_MIN"] = env
    w = ex.words.clone()
    out[name] = {"sub_ms": tm(lambda: ex - ey), "pow2_52_ms": tm(lambda: h.ct_pow2_(w, d52)),
                 "pow2_12_ms": tm(lambda: h.ct_pow2_(w, d12))}
    r = (ex - ey)
    out[name]["checksum"] = int(r.words.to(torch.int64).sum().item())
print(json.dumps(out))
