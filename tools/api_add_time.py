"""Dev probe: API-level a + b / a - b on 2^20 random floats (exponents differ per element), with and without the
|delta|-sorted pass of paillier._add_aligned (PAI_ALIGN_SORT_MIN)."""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
from bench import synthetic_key
from pailliercryptolib_python_amd import PaillierPrivateKey, PaillierPublicKey
from pailliercryptolib_python_amd.bindings import ipclPublicKey
key = synthetic_key(2048, 0x1234567)
pk = PaillierPublicKey(ipclPublicKey(key.n, 2048, True, hs=key.hs, randbits=key.randbits))
sk = PaillierPrivateKey(pk, key.p, key.q)
B = 1 << 20
rng = np.random.default_rng(1)
x, y = rng.uniform(-1000, 1000, B), rng.uniform(-1000, 1000, B)
a, b = pk.encrypt(x), pk.encrypt(y)
def tm(f, reps=3):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return round((time.perf_counter() - t0) / reps * 1e3, 2)
row = {"batch": B}
for tag, env in (("sorted", "1"), ("unsorted", str(1 << 40))):
    os.environ["PAI_ALIGN_SORT_MIN"] = env
    row[f"a_plus_b_{tag}_ms"] = tm(lambda: a + b)
    row[f"a_minus_b_{tag}_ms"] = tm(lambda: a - b, reps=2)
    row[f"a_plus_plain_{tag}_ms"] = tm(lambda: a + y)
os.environ.pop("PAI_ALIGN_SORT_MIN")
s = a + b
assert np.allclose(sk.decrypt_to_numpy(s), x + y, rtol=0, atol=1e-6)
print(json.dumps(row))
