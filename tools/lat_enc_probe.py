"""Dev probe: small-batch DJN encrypt latency, wave-shared chains (k_encrypt_tree, default) against one chain per integer (PAI_TUNE=lat_enc_tree=0)."""
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
for N in (16, 256, 1024, 4096):
    m = torch.randint(0, 2**31 - 1, (N, pub.n_words), dtype=torch.int64, device=dev, generator=g).to(torch.int32)
    m[:, -1] &= 0x0FFFFFFF
    r = pub.random_r(N, generator=g)
    row = {"bits": bits, "N": N}
    ref = None
    for tree, m1 in (("1000000", "1"), ("1000000", "0"), ("0", "1")):
        os.environ["PAI_TUNE"] = f"lat_enc_tree={tree}"
        os.environ["PAI_DISABLE"] = "lat_enc_m1" if m1 == "0" else ""
        ct = pub.encrypt(m, r)
        if ref is None: ref = ct.clone()
        assert torch.equal(ct, ref), (N, tree, m1)
        row[f"enc_tree{'1' if tree != '0' else '0'}_m1{m1}_ms"] = round(tm(lambda: pub.encrypt(m, r)), 3)
    print(json.dumps(row), flush=True)
