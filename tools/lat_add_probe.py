"""Dev probe: small-batch ct+ct (wire form, single tagged product, aligned with delta <= 13) on the latency geometry (PAI_LAT_ADD_MAX) against the throughput geometry."""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from bench import synthetic_key
from pailliercryptolib_python_amd import engine
dev = torch.device('cuda', 0)
bits = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
key = synthetic_key(bits, 0x1234567)
pub = engine.PublicKeyHandle(key.n, bits, key.hs, key.randbits, device=dev)
def tm(f, reps=10):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
g = torch.Generator(device=dev); g.manual_seed(1)
for N in (16, 64, 256, 1024, 2048, 4096, 8192):
    m = torch.randint(0, 2**31 - 1, (N, pub.n_words), dtype=torch.int64, device=dev, generator=g).to(torch.int32)
    m[:, -1] &= 0x0FFFFFFF
    a = pub.encrypt(m, pub.random_r(N, generator=g)); b = pub.encrypt(m, pub.random_r(N, generator=g))
    delta = (torch.arange(N, device=dev) % 14).to(torch.int32)
    row = {"bits": bits, "N": N}
    ref = None
    for lat in ("1000000", "m1off", "0"):
        os.environ["PAI_LAT_ADD_MAX"] = "1000000" if lat == "m1off" else lat
        if lat == "m1off": os.environ["PAI_DISABLE"] = "lat_add_m1"          # the conventional context of n^2 (before round 6)
        else: os.environ.pop("PAI_DISABLE", None)
        outs = (pub.ct_add(a, b), pub.ct_mont_mul(a, b), pub.ct_add_aligned(a, b, delta))
        if ref is None: ref = [o.clone() for o in outs]
        assert all(torch.equal(o, r) for o, r in zip(outs, ref)), (N, lat)
        k = {"1000000": "lat", "m1off": "lat_conv", "0": "thr"}[lat]
        row[f"add_{k}_ms"] = round(tm(lambda: pub.ct_add(a, b)), 4)
        row[f"mont_{k}_ms"] = round(tm(lambda: pub.ct_mont_mul(a, b)), 4)
        row[f"aligned13_{k}_ms"] = round(tm(lambda: pub.ct_add_aligned(a, b, delta)), 4)
        d2 = (delta % 3).contiguous()
        row[f"aligned2_{k}_ms"] = round(tm(lambda: pub.ct_add_aligned(a, b, d2)), 4)
    print(json.dumps(row), flush=True)
