"""Dev probe: API-level latency at the reference's benchmark sizes (bench/bench_ipcl_python.py:24-25,34-35,45-46)."""
import os, sys, time, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
from bench import synthetic_key
from pailliercryptolib_python_amd import PaillierPrivateKey, PaillierPublicKey
from pailliercryptolib_python_amd.bindings import ipclPublicKey
key = synthetic_key(2048, 0x1234567)
pk = PaillierPublicKey(ipclPublicKey(key.n, 2048, True, hs=key.hs, randbits=key.randbits))
sk = PaillierPrivateKey(pk, key.p, key.q)
def tm(f, reps=5):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return round((time.perf_counter() - t0) / reps * 1e3, 3)
for n in (16, 64):
    x = np.random.default_rng(n).uniform(-100, 100, n)
    y = np.random.default_rng(n + 1).uniform(-100, 100, n)
    ex, ey = pk.encrypt(x), pk.encrypt(y)
    row = {"n": n,
           "encrypt_ms": tm(lambda: pk.encrypt(x)), "decrypt_ms": tm(lambda: sk.decrypt(ex)),
           "add_ctct_ms": tm(lambda: ex + ey), "add_ctpt_ms": tm(lambda: ex + y), "mul_ctpt_ms": tm(lambda: ex * y),
           "sum_ms": tm(lambda: ex.sum())}
    pk.precompute_obfuscators(6 * n)
    row["encrypt_pooled_ms"] = tm(lambda: pk.encrypt(x))
    print(json.dumps(row))
