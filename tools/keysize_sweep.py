#!/usr/bin/env python3
"""Kernel times of encrypt / decrypt / ct+ct / ct*pt for the BASELINE key sizes on one GPU (HIP events on
the launch stream, through the same engine handles bench.py uses).  One JSON line per key size.

    python tools/keysize_sweep.py [--batch 65536] [--bits 1024 2048 3072 4096]
"""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from bench import synthetic_key                        # noqa: E402
from pailliercryptolib_python_amd import engine, fixedpoint  # noqa: E402


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--bits", type=int, nargs="*", default=[1024, 2048, 3072, 4096])
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    for bits in args.bits:
        key = synthetic_key(bits, 0x1234567)
        p, q = key.p, key.q
        t0 = time.perf_counter()
        pub = engine.PublicKeyHandle(key.n, bits, key.hs, key.randbits, device=dev)
        priv = engine.PrivateKeyHandle(pub, p, q)
        torch.cuda.synchronize()
        t_key = time.perf_counter() - t0
        B = args.batch
        x = np.random.default_rng(7).uniform(-1000.0, 1000.0, B)
        res, _ = fixedpoint.encode_float64_array(x, key.n, pub.n_words)
        m = engine.to_device_words(res, dev)
        gen = torch.Generator(device=dev)
        gen.manual_seed(11)
        r = pub.random_r(B, generator=gen)
        ct, out = pub.empty_ct(B), pub.empty_pt(B)
        times = {}
        engine.profile_enable(True)
        for rep in range(2):
            pub.encrypt(m, r, out=ct)
            t_enc = dict(engine.profile_last())
            priv.decrypt(ct, out=out)
            t_dec = dict(engine.profile_last())
        ok = bool(torch.equal(out, m))
        ct2 = pub.empty_ct(B)

        def wall(f):
            f()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            f()
            torch.cuda.synchronize()
            return {"wall": (time.perf_counter() - t1) * 1e3}

        t_add = wall(lambda: pub.ct_add(ct, ct, out=ct2))
        # 53-bit multipliers (what a float mantissa encodes to)
        e = torch.randint(1, 1 << 30, (B, 2), dtype=torch.int32, device=dev)
        e[:, 1] &= (1 << 21) - 1
        t_mul = wall(lambda: pub.ct_mul(ct, e, 53, out=ct2))
        t_inv = wall(lambda: pub.ct_invert(ct, out=ct2))
        engine.profile_enable(False)
        times = {"encrypt_ms": sum(t_enc.values()), "decrypt_ms": sum(t_dec.values()), "ct_add_ms": sum(t_add.values()),
                 "ct_mul53_ms": sum(t_mul.values()), "ct_invert_ms": sum(t_inv.values())}
        print(json.dumps({"key_bits": bits, "batch": B, "roundtrip_ok": ok, "key_setup_s": round(t_key, 3),
                          **{k: round(v, 3) for k, v in times.items()},
                          "enc_dec_ops_per_s": round(B / ((times["encrypt_ms"] + times["decrypt_ms"]) * 1e-3)),
                          "kernels": {**t_enc, **t_dec}}), flush=True)
        del pub, priv


main()
