"""ct+ct timing A/B over library variants (PAI_NATIVE_LIB): wire-form add, in-chain add (one product), broadcast add, with a
bit check of each against CPython on a few rows.  One JSON line."""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
from bench import synthetic_key
from pailliercryptolib_python_amd import engine, fixedpoint
dev = torch.device('cuda', 0)
bits = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
key = synthetic_key(bits, 0x1234567)
pub = engine.PublicKeyHandle(key.n, bits, key.hs, key.randbits, device=dev)
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
g = torch.Generator(device=dev); g.manual_seed(5)
ct = torch.randint(-2**31, 2**31 - 1, (B, pub.ct_words), dtype=torch.int32, device=dev, generator=g)
ct[:, -1] &= 0x3FFFFFFF                       # residues below n^2 (top two bits clear)
ct_b = torch.roll(ct, 3, dims=0).contiguous()
out = pub.empty_ct(B)
def tm(f, reps=5):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
res = {"lib": os.environ.get("PAI_NATIVE_LIB", "default"), "bits": bits, "batch": B}
idx = [0, 1, B // 2, B - 1]
def rows(t): return engine.words_to_ints(engine.to_host_words(t[idx]))
a_h, b_h = rows(ct), rows(ct_b)
rinv = pow(pow(2, pub.mont_bits, key.nsq), -1, key.nsq)
res["add_ms"] = tm(lambda: pub.ct_add(ct, ct_b, out=out)); ok = rows(out) == [a * b % key.nsq for a, b in zip(a_h, b_h)]
res["mont_ms"] = tm(lambda: pub.ct_mont_mul(ct, ct_b, out=out)); ok = ok and rows(out) == [a * b * rinv % key.nsq for a, b in zip(a_h, b_h)]
res["bcast_ms"] = tm(lambda: pub.ct_add(ct, ct_b[:1], out=out)); ok = ok and rows(out) == [a * engine.words_to_ints(engine.to_host_words(ct_b[:1]))[0] % key.nsq for a in a_h]
res["inplace_ms"] = tm(lambda: pub.ct_mont_mul(out, ct_b, out=out))
res["ok"] = bool(ok)
res["mont_Mops"] = B / res["mont_ms"] / 1e3
print(json.dumps(res))
