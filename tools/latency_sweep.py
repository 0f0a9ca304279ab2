"""Dev probe: decrypt / encrypt / ct*pt latency over the batch size with the library's own switches (def), the latency paths forced
(PAI_LATENCY_MAX huge) and the throughput kernels only (PAI_LATENCY_MAX = 0).  A second argument selects the dense grid."""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
from bench import synthetic_key
from pailliercryptolib_python_amd import engine
dev = torch.device('cuda', 0)
bits = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
key = synthetic_key(bits, 0x1234567)
pub = engine.PublicKeyHandle(key.n, bits, key.hs, key.randbits, device=dev)
priv = engine.PrivateKeyHandle(pub, key.p, key.q)
def tm(f, reps=3):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
g = torch.Generator(device=dev); g.manual_seed(1)
GRID = (1, 16, 64, 128, 256, 384, 512, 768, 1024, 1536, 2048, 3072, 4096, 6144, 8192, 12288, 16384, 32768) if len(sys.argv) > 2 else (1, 16, 64, 256, 512, 1024, 2048, 4096, 16384)
for N in GRID:
    m = torch.randint(0, 2**31 - 1, (N, pub.n_words), dtype=torch.int64, device=dev, generator=g).to(torch.int32)
    m[:, -1] &= 0x0FFFFFFF
    r = pub.random_r(N, generator=g)
    ct = pub.encrypt(m, r)
    row = {"bits": bits, "N": N}
    for name, sw in (("def", None), ("lat", "1000000"), ("thr", "0")):       # def: the library's own switches
        if sw is None: os.environ.pop("PAI_LATENCY_MAX", None)
        else: os.environ["PAI_LATENCY_MAX"] = sw
        out = priv.decrypt(ct)
        assert torch.equal(out, m), (N, name)
        row[f"dec_{name}_ms"] = round(tm(lambda: priv.decrypt(ct)), 3)
        assert torch.equal(pub.encrypt(m, r), ct), (N, name, "encrypt")
        row[f"enc_{name}_ms"] = round(tm(lambda: pub.encrypt(m, r)), 3)
        e = torch.randint(1, 1 << 30, (N, 2), dtype=torch.int32, device=dev); e[:, 1] &= (1 << 21) - 1
        row[f"mul_{name}_ms"] = round(tm(lambda: pub.ct_mul(ct, e, 53)), 3)
    print(json.dumps(row), flush=True)
