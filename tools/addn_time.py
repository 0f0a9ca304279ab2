#!/usr/bin/env python3
"""Times pai_ct_addn (n-ary ciphertext sum in one launch) against the chain of single products it replaces.
usage: python tools/addn_time.py [--bits 2048] [--batch 1048576] [--k 8]   -> one JSON line"""
import argparse, json, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch
import bench
from pailliercryptolib_python_amd import engine

ap = argparse.ArgumentParser()
ap.add_argument("--bits", type=int, default=2048)
ap.add_argument("--batch", type=int, default=1 << 20)
ap.add_argument("--k", type=int, default=8)
ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
key = bench.synthetic_key(a.bits)
dev = torch.device("cuda", 0)
pub = engine.PublicKeyHandle(key.n, a.bits, key.hs, key.randbits, device=dev)
N, k = a.batch, a.k
g = torch.Generator(device=dev); g.manual_seed(1)
ops = []
for j in range(k):
    t = torch.randint(-(1 << 31), 1 << 31, (N, pub.ct_words), dtype=torch.int64, device=dev, generator=g).to(torch.int32)
    t[:, -1] &= 0x3FFFFFFF          # below n^2 for the fixture keys (top word of n^2 is large)
    ops.append(t.contiguous())
out = pub.empty_ct(N)

def wall(f, reps):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps

def chain():
    acc = pub.ct_mont_mul(ops[0], ops[1])
    for j in range(2, k): pub.ct_mont_mul(acc, ops[j], out=acc)
    return acc

t_addn = wall(lambda: pub.ct_addn(ops, None, 0, 0, 1 - k, out=out), a.reps)
t_addn_wire = wall(lambda: pub.ct_addn(ops, None, 0, 0, 0, out=out), a.reps)
t_chain = wall(chain, a.reps)
same = bool(torch.equal(pub.ct_addn(ops, None, 0, 0, 1 - k), chain()))
idx = [0, 1, N // 2, N - 1]
got = engine.words_to_ints(engine.to_host_words(pub.ct_addn(ops, None, 0, 0, 0)[idx]))
want = []
for i in idx:
    w = 1
    for t in ops: w = w * engine.words_to_ints(engine.to_host_words(t[i:i + 1]))[0] % key.nsq
    want.append(w)
NL = bench.modmul_limbs(2 * a.bits)
macs = 2 * NL * NL * (k - 1)
print(json.dumps({"bits": a.bits, "batch": N, "k": k, "addn_ms": 1e3 * t_addn, "addn_wire_ms": 1e3 * t_addn_wire, "chain_ms": 1e3 * t_chain,
                  "ms_per_product": 1e3 * t_addn / (k - 1), "executed_frac": macs * N / t_addn / bench.PEAK_MAC32_PER_S,
                  "hbm_GBs": (k + 1) * N * pub.ct_words * 4 / t_addn / 1e9, "same_bits_as_chain": same, "oracle_ok": got == want}))
