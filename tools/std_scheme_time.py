"""Times the standard-scheme (non-DJN) encryption r^n path (k_pow_padic at <= 2048-bit keys) and apply_obfuscator
through the public API: python tools/std_scheme_time.py [bits] [batch]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pailliercryptolib_python_amd import PaillierKeypair  # noqa: E402

bits = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
N = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
x = np.random.default_rng(1).uniform(-100, 100, N)
out = {"key_bits": bits, "batch": N}
for djn in (False, True):
    pk, sk = PaillierKeypair.generate_keypair(bits, djn)
    ct = pk.encrypt(x)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        ct = pk.encrypt(x)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    out["encrypt_ms_djn" if djn else "encrypt_ms_standard"] = round(best * 1e3, 2)
    back = sk.decrypt_to_numpy(ct)
    out["ok_djn" if djn else "ok_standard"] = bool(np.array_equal(back, x))
print(json.dumps(out))
