"""Dev probe: API-level operations on 2^20 elements (mixed exponents, mixed signs)."""
import os, sys, time, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
from bench import synthetic_key
from pailliercryptolib_python_amd import PaillierPrivateKey, PaillierPublicKey
from pailliercryptolib_python_amd.bindings import ipclPublicKey
key = synthetic_key(2048, 0x1234567)
pk = PaillierPublicKey(ipclPublicKey(key.n, 2048, True, hs=key.hs, randbits=key.randbits))
sk = PaillierPrivateKey(pk, key.p, key.q)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
rng = np.random.default_rng(1)
x, y, w = rng.uniform(-1000, 1000, B), rng.uniform(-1000, 1000, B), rng.uniform(-10, 10, B)
def tm(f, reps=2):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): r = f()
    torch.cuda.synchronize()
    return round((time.perf_counter() - t0) / reps * 1e3, 1)
ex, ey = pk.encrypt(x), pk.encrypt(y)
row = {"B": B, "encrypt": tm(lambda: pk.encrypt(x)), "add_ctct": tm(lambda: ex + ey), "add_ctpt": tm(lambda: ex + y),
       "mul_ctpt_mixed_sign": tm(lambda: ex * w), "mul_scalar": tm(lambda: ex * 2.5), "sub_ctct": tm(lambda: ex - ey),
       "sum": tm(lambda: ex.sum()), "decrypt_np": tm(lambda: sk.decrypt_to_numpy(ex))}
s = ex + ey
assert np.allclose(sk.decrypt_to_numpy(s), x + y)
p = ex * w
assert np.allclose(sk.decrypt_to_numpy(p), x * w, rtol=1e-9, atol=1e-9)
print(json.dumps(row))
