"""Dev probe: small-batch raw encryption (1 + m n) and the pieces of a + plaintext at the API level."""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
from bench import synthetic_key
from pailliercryptolib_python_amd import engine
dev = torch.device('cuda', 0)
key = synthetic_key(2048, 0x1234567)
pub = engine.PublicKeyHandle(key.n, 2048, key.hs, key.randbits, device=dev)
def tm(f, reps=20):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): f(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6
g = torch.Generator(device=dev); g.manual_seed(1)
engine.profile_enable(True)
for N in (16, 64, 1024):
    m = torch.randint(0, 2**31 - 1, (N, pub.n_words), dtype=torch.int64, device=dev, generator=g).to(torch.int32)
    m[:, -1] &= 0x0FFFFFFF
    x = torch.rand(N, dtype=torch.float64, device=dev) * 1000
    xh = x.cpu().numpy()
    row = {"N": N}
    row["raw_encrypt_us"] = round(tm(lambda: pub.raw_encrypt(m)), 1); row["raw_kernel"] = engine.profile_last()
    row["fp_encode_us"] = round(tm(lambda: pub.fp_encode_f64(x)), 1)
    row["h2d_us"] = round(tm(lambda: torch.from_numpy(xh).to(dev)), 1)
    print(json.dumps(row), flush=True)
