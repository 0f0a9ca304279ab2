import os, sys, time, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
from bench import synthetic_key
from pailliercryptolib_python_amd import PaillierPrivateKey, PaillierPublicKey, engine
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
d = torch.from_numpy((ex._expo.astype(np.int64) - ey._expo.astype(np.int64)).astype(np.int32)).to(h.device)
print("delta stats", int(d.abs().max()), float(d.abs().float().mean()))
out = h.empty_ct(B)
print("add_aligned kernel only", tm(lambda: h.ct_add_aligned(ex.words, ey.words, d, out=out)))
print("add_aligned alloc out", tm(lambda: h.ct_add_aligned(ex.words, ey.words, d)))
print("ct_add", tm(lambda: h.ct_add(ex.words, ey.words, out=out)))
print("raw encrypt API", tm(lambda: pk.encrypt(y, apply_obfuscator=False)))
ry = pk.encrypt(y, apply_obfuscator=False)
print("ex + ry", tm(lambda: ex + ry))
print("ex + ey", tm(lambda: ex + ey))
print("ex + y", tm(lambda: ex + y))
engine.profile_enable(True)
h.ct_add_aligned(ex.words, ey.words, d, out=out); print(engine.profile_last())
m = torch.zeros((B, h.n_words), dtype=torch.int32, device=h.device)
h.raw_encrypt(m); print(engine.profile_last())
