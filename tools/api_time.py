#!/usr/bin/env python3
"""Wall-clock breakdown of the PUBLIC API on host arrays (numpy in, numpy/list out), 2048-bit DJN key:
what a user of the reference sees, PCIe and host codec included.  One JSON line."""
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from pailliercryptolib_python_amd import fixedpoint as fp  # noqa: E402


def t(f, *a, **k):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = f(*a, **k)
    torch.cuda.synchronize()
    return r, time.perf_counter() - t0


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
    from pailliercryptolib_python_amd import PaillierKeypair
    pk, sk = PaillierKeypair.generate_keypair(2048, True)
    x = np.random.default_rng(3).uniform(-1000, 1000, N)
    out = {"N": N}
    ct, out["encrypt_s"] = t(pk.encrypt, x)
    ct, out["encrypt_again_s"] = t(pk.encrypt, x)
    h = pk.pubkey.handle
    (_, _), out["  codec_encode_s"] = t(fp.encode_array, x, pk.n, pk.max_int, h.n_words)
    _, out["  draw_r_s"] = t(pk.pubkey._draw_r, N)
    s, out["add_s"] = t(lambda: ct + ct)
    m, out["mul_scalar_s"] = t(lambda: ct * 3.5)
    y, out["decrypt_to_numpy_first_s"] = t(sk.decrypt_to_numpy, ct)       # first call: grows the key's device scratch
    y, out["decrypt_to_numpy_s"] = t(sk.decrypt_to_numpy, ct)
    out["roundtrip_ok"] = bool(np.array_equal(y, x))
    if N <= (1 << 18):
        z, out["decrypt_list_s"] = t(sk.decrypt, s)
        out["add_ok"] = bool(np.allclose(np.asarray(z, dtype=np.float64), 2 * x, rtol=0, atol=1e-9))
    out = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in out.items()}
    out["enc_dec_ops_per_s_api"] = round(N / (out["encrypt_again_s"] + out["decrypt_to_numpy_s"]))
    print(json.dumps(out))


main()
