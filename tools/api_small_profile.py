"""Dev probe: where the microseconds of a small public-API call go (cProfile over 2000 calls each of a + b, a + pt, a * pt, encrypt, decrypt at 16 elements)."""
import cProfile, pstats, io, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
from bench import synthetic_key
from pailliercryptolib_python_amd import PaillierKeypair, PaillierPublicKey, PaillierPrivateKey
key = synthetic_key(2048, 0x1234567)
pk = PaillierPublicKey(key.n, 2048, True, hs=key.hs, randbits=key.randbits) if 'hs' in PaillierPublicKey.__init__.__code__.co_varnames else None
if pk is None:
    pk, sk = PaillierKeypair.generate_keypair(2048)
else:
    sk = PaillierPrivateKey(pk, key.p, key.q)
x = (np.arange(16) + 11) * 1234.5678
y = (np.arange(16) + 3) * 0.4321
a, b = pk.encrypt(x), pk.encrypt(y)
def timeit(name, f, reps=300):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    print(name, round((time.perf_counter() - t0) / reps * 1e6, 1), "us")
for name, f in (("add_ctct", lambda: a + b), ("add_ctpt", lambda: a + y), ("mul_ctpt", lambda: a * y), ("encrypt", lambda: pk.encrypt(x)), ("decrypt", lambda: sk.decrypt(a))):
    timeit(name, f)
for name, f in (("add_ctct", lambda: a + b), ("encrypt", lambda: pk.encrypt(x))):
    pr = cProfile.Profile(); pr.enable()
    for _ in range(500): f()
    torch.cuda.synchronize(); pr.disable()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(18); print(name); print(s.getvalue()[:3500])
