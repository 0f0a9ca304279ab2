"""Dev probe: decrypt latency of the smallest batches: four-wave digit-pair pipeline (k_dec_a_pp, default) against the
wave-pair right-to-left kernel (PAI_TUNE=lat_pp=0) and the window kernel (lat_pp=0,lat_rl=0).   python tools/lat_pp_probe.py [bits]"""
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
for N in ((1, 16, 64, 128) if len(sys.argv) < 3 else (16, 128, 256, 384, 512, 640, 768, 1024, 1280)):
    m = torch.randint(0, 2**31 - 1, (N, pub.n_words), dtype=torch.int64, device=dev, generator=g).to(torch.int32)
    m[:, -1] &= 0x0FFFFFFF
    ct = pub.encrypt(m, pub.random_r(N, generator=g))
    row = {"bits": bits, "N": N}
    for name, tune in (("pp", "lat_pp=100000"), ("rl", "lat_pp=0"), ("window", "lat_pp=0,lat_rl=0")) + ((("default", ""),) if len(sys.argv) > 2 else ()):
        os.environ["PAI_TUNE"] = tune
        ok = bool(torch.equal(priv.decrypt(ct), m))
        engine.profile_enable(True)
        priv.decrypt(ct)
        k = engine.profile_last().get("k_dec_a")
        engine.profile_enable(False)
        row[name] = {"ok": ok, "wall_ms": round(tm(lambda: priv.decrypt(ct)), 3), "k_dec_a_ms": round(k, 3) if k else None}
    print(json.dumps(row), flush=True)
