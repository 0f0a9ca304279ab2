"""Dev probe: the public API's `a + b` on two encrypted float arrays (uniform in [-1000, 1000]: 61 % of the element pairs share their
exponent, the rest differ by 4 .. 12 bits) — wall time per call at 65 536 and 2^20 elements, and the split of the exponent
differences.   python tools/api_add_float_time.py"""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import torch
from bench import synthetic_key
from pailliercryptolib_python_amd import PaillierPublicKey, PaillierPrivateKey
from pailliercryptolib_python_amd.bindings import ipclPublicKey

key = synthetic_key(2048, 0x1234567)
pk = PaillierPublicKey(ipclPublicKey(key.n, 2048, True, hs=key.hs, randbits=key.randbits))
sk = PaillierPrivateKey(pk, key.p, key.q)
for lg in (16, 20):
    N = 1 << lg
    rng = np.random.default_rng(lg)
    a, b = rng.uniform(-1000, 1000, N), rng.uniform(-1000, 1000, N)
    ea, eb = pk.encrypt(a, apply_obfuscator=False), pk.encrypt(b, apply_obfuscator=False)
    d = np.asarray(ea.exponent()) - np.asarray(eb.exponent())
    s = ea + eb; s.ciphertext().words; torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        s = ea + eb
        s.ciphertext().words                      # the wire form (what a decryption or an export would ask for)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    row = {"elements": N, "ms_per_add": round(dt * 1e3, 3), "M_per_s": round(N / dt / 1e6, 2),
           "delta_zero_share": round(float((d == 0).mean()), 3), "delta_abs_max": int(np.abs(d).max())}
    idx = np.linspace(0, N - 1, 64).astype(int)
    got = sk.decrypt(s)
    row["max_abs_err_sample"] = float(np.max(np.abs(np.asarray(got)[idx] - (a + b)[idx])))
    print(json.dumps(row), flush=True)
