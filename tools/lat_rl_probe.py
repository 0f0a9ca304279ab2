"""Dev probe: small-batch decrypt latency, right-to-left wave pairs (default; PAI_TUNE=lat_rl=0 switches them off) against the left-to-right window kernel."""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from bench import synthetic_key
from pailliercryptolib_python_amd import engine
dev = torch.device('cuda', 0)
bits = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
key = synthetic_key(bits, 0x1234567)
pub = engine.PublicKeyHandle(key.n, bits, key.hs, key.randbits, device=dev)
priv = engine.PrivateKeyHandle(pub, key.p, key.q)
def tm(f, reps=5):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
g = torch.Generator(device=dev); g.manual_seed(1)
for N in (16, 256, 257, 384, 512, 513, 768, 1024, 2048, 4096):
    m = torch.randint(0, 2**31 - 1, (N, pub.n_words), dtype=torch.int64, device=dev, generator=g).to(torch.int32)
    m[:, -1] &= 0x0FFFFFFF
    ct = pub.encrypt(m, pub.random_r(N, generator=g))
    row = {"bits": bits, "N": N}
    for rl in ("100000", "0"):
        os.environ["PAI_TUNE"] = f"lat_rl={rl}"
        assert torch.equal(priv.decrypt(ct), m), (N, rl)
        row[f"dec_rl{'1' if rl != '0' else '0'}_ms"] = round(tm(lambda: priv.decrypt(ct)), 3)
    print(json.dumps(row), flush=True)
