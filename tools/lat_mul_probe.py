"""Dev probe: small-batch ct*pt latency (dense 53-bit exponents), the four-wave digit-pair pipeline (k_ctmul_pp) against right-to-left wave pairs (k_modexp_rl, PAI_TUNE=lat_mul_pp=0) and the windowed kernel (lat_mul_rl=0 too)."""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from bench import synthetic_key
from pailliercryptolib_python_amd import engine
dev = torch.device('cuda', 0)
bits = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
key = synthetic_key(bits, 0x1234567)
pub = engine.PublicKeyHandle(key.n, bits, key.hs, key.randbits, device=dev)
def tm(f, reps=5):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
g = torch.Generator(device=dev); g.manual_seed(1)
for N in (1, 16, 64, 256, 512, 1024, 2048, 4096):
    m = torch.randint(0, 2**31 - 1, (N, pub.n_words), dtype=torch.int64, device=dev, generator=g).to(torch.int32)
    m[:, -1] &= 0x0FFFFFFF
    ct = pub.encrypt(m, pub.random_r(N, generator=g))
    e = torch.randint(-2**31, 2**31 - 1, (N, 2), dtype=torch.int64, device=dev, generator=g).to(torch.int32)
    e[:, 1] &= (1 << 21) - 1
    e[:, 1] |= 1 << 20
    row = {"bits": bits, "N": N}
    ref = None
    for name, knobs in (("pp", "lat_mul_pp=1000000"), ("rl", "lat_mul_pp=0,lat_mul_rl=1000000"), ("win", "lat_mul_pp=0,lat_mul_rl=0")):
        os.environ["PAI_TUNE"] = knobs
        out = pub.ct_mul(ct, e, 53)
        if ref is None: ref = out.clone()
        assert torch.equal(out, ref), (N, name)
        row[f"mul_{name}_ms"] = round(tm(lambda: pub.ct_mul(ct, e, 53)), 3)
    print(json.dumps(row), flush=True)
