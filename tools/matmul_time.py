"""Encrypted matrix x plaintext matrix through pai_ct_multiexp (default for >= 2^19 terms) and term by term
(PAI_MEXP_MIN_TERMS=huge):  python tools/matmul_time.py [m n k [key_bits]]"""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
from bench import synthetic_key
from pailliercryptolib_python_amd import PaillierPublicKey, engine
from pailliercryptolib_python_amd.bindings import ipclPublicKey
m, n, k = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (64, 1024, 64)
bits = int(sys.argv[4]) if len(sys.argv) > 4 else 2048
if bits == 2048:
    key = synthetic_key(2048, 0x1234567)
else:
    from tests.test_gpu_paillier_abi import seeded_key
    key = seeded_key(bits)
pk = PaillierPublicKey(ipclPublicKey(key.n, bits, True, hs=key.hs, randbits=key.randbits))
rng = np.random.default_rng(3)
x = rng.uniform(-10, 10, m * n)
w = rng.standard_normal((n, k))
en = pk.encrypt(x)
def tm(f, reps=2):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
out = {"key_bits": bits, "m": m, "n": n, "k": k, "terms": m * n * k}
for name, env in (("multiexp", None), ("term_by_term", str(1 << 60))):
    if env is None: os.environ.pop("PAI_MEXP_MIN_TERMS", None)
    else: os.environ["PAI_MEXP_MIN_TERMS"] = env
    t = tm(lambda: en @ w)
    r = en @ w
    out[name] = {"s": round(t, 4), "terms_per_s": round(m * n * k / t), "checksum": int(r.words.to(torch.int64).sum().item()),
                 "expo_sum": int(np.asarray(r.exponent()).astype(np.int64).sum())}
    if env is None:
        import pailliercryptolib_python_amd.paillier as _pm
        torch.cuda.synchronize(); t0 = time.perf_counter()
        _orig = engine.PublicKeyHandle.ct_multiexp
        tt = {}
        def _timed(self, *a, **kw):
            torch.cuda.synchronize(); t1 = time.perf_counter()
            r_ = _orig(self, *a, **kw); torch.cuda.synchronize(); tt["ct_multiexp_s"] = round(time.perf_counter() - t1, 4); return r_
        engine.PublicKeyHandle.ct_multiexp = _timed
        en @ w; torch.cuda.synchronize(); tt["total_s"] = round(time.perf_counter() - t0, 4)
        engine.PublicKeyHandle.ct_multiexp = _orig
        out[name]["split"] = tt
        engine.profile_enable(True); en @ w; out[name]["kernels_ms"] = {a: round(b, 2) for a, b in engine.profile_last().items()}; engine.profile_enable(False)
ND = (1 << 20) if bits <= 2048 else (1 << 17)
big = pk.encrypt(rng.uniform(-10, 10, ND))
v = rng.standard_normal(ND)
for name, env in (("dot_multiexp", None), ("dot_term_by_term", str(1 << 60))):
    if env is None: os.environ.pop("PAI_MEXP_MIN_TERMS", None)
    else: os.environ["PAI_MEXP_MIN_TERMS"] = env
    t = tm(lambda: big.dot(v))
    r = big.dot(v)
    out[name] = {"s": round(t, 4), "checksum": int(r.words.to(torch.int64).sum().item())}
print(json.dumps(out))
