import sys, time, json
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
from bench import synthetic_key
from pailliercryptolib_python_amd import engine, fixedpoint
dev = torch.device('cuda', 0)
key = synthetic_key(2048, 0x1234567)
pub = engine.PublicKeyHandle(key.n, 2048, key.hs, key.randbits, device=dev)
B = 1 << 20
x = np.random.default_rng(7).uniform(-1000.0, 1000.0, B)
res, _ = fixedpoint.encode_float64_array(x, key.n, pub.n_words)
m = engine.to_device_words(res, dev)
gen = torch.Generator(device=dev); gen.manual_seed(11)
r = pub.random_r(B, generator=gen)
ct = pub.encrypt(m, r)
ct2 = pub.empty_ct(B)
def tm(f, reps=3):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
out = {}
out['ct_add_ms'] = tm(lambda: pub.ct_add(ct, ct, out=ct2))
out['ct_mont_mul_ms'] = tm(lambda: pub.ct_mont_mul(ct, ct, out=ct2))          # the addition inside a chain (lazy Montgomery domain)
e = torch.randint(0, 1 << 30, (B, 2), dtype=torch.int32, device=dev); e[:, 1] &= (1 << 21) - 1; e[:, 1] |= (1 << 20)
out['ct_mul_53bit_ms'] = tm(lambda: pub.ct_mul(ct, e, 53, out=ct2))
e1 = e[:1].contiguous()
out['ct_mul_53bit_bcast_ms'] = tm(lambda: pub.ct_mul(ct, e1, 53, out=ct2))
d = torch.full((B,), 12, dtype=torch.int32, device=dev)
c3 = ct.clone()
out['ct_pow2_12_ms'] = tm(lambda: pub.ct_pow2_(c3, d))
out['ct_invert_ms'] = tm(lambda: pub.ct_invert(ct, out=ct2))
print(json.dumps({k: round(v, 2) for k, v in out.items()}))
