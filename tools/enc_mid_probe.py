"""Dev probe: DJN encryption (and re-obfuscation) of mid-size batches at keys up to 2048 bits — the lane-group digit-pair kernel with 4
lanes per element reading the one-element-per-lane engine's fixed-base table (PAI_TUNE=enc_mid_min=0,enc_mid_max=huge) against the
library's other paths; identical ciphertext bits required.   python tools/enc_mid_probe.py [bits]"""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from bench import synthetic_key
from pailliercryptolib_python_amd import engine
dev = torch.device('cuda', 0)
bits = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
if bits in (1024, 2048, 3072, 4096):
    key = synthetic_key(bits, 0x1234567)
else:                                                    # other sizes: the native generator, seeded
    from types import SimpleNamespace
    from pailliercryptolib_python_amd import _native
    p_, q_ = sorted(_native.keygen(bits, True, seed=bits))
    n_ = p_ * q_
    key = SimpleNamespace(bits=bits, p=p_, q=q_, n=n_, nsq=n_ * n_, hs=pow((-0x1234567 ** 2) % (n_ * n_), n_, n_ * n_), randbits=bits // 2)
pub = engine.PublicKeyHandle(key.n, bits, key.hs, key.randbits, device=dev)
def tm(f, reps=3):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
g = torch.Generator(device=dev); g.manual_seed(1)
for N in ((1024, 2048, 4096, 6144, 8192, 12288, 16384, 24576, 32768, 49152, 65536, 131072) if bits in (1024, 2048) else (2048, 4096, 8192, 16384, 32768, 49152)):
    m = torch.randint(0, 2**31 - 1, (N, pub.n_words), dtype=torch.int64, device=dev, generator=g).to(torch.int32)
    m[:, -1] &= 0x0FFFFFFF
    r = pub.random_r(N, generator=g)
    row = {"bits": bits, "N": N}
    ref = ref2 = None
    for name, tune in (("other", "enc_mid_max=0"), ("mid", "enc_mid_min=0,enc_mid_max=100000000")):
        os.environ["PAI_TUNE"] = tune
        ct = pub.encrypt(m, r)
        ob = pub.obfuscate_(ct.clone(), r)
        if ref is None: ref, ref2 = ct.clone(), ob.clone()
        row[name] = {"same": bool(torch.equal(ct, ref)) and bool(torch.equal(ob, ref2)), "ms": round(tm(lambda: pub.encrypt(m, r)), 3)}
    print(json.dumps(row), flush=True)
