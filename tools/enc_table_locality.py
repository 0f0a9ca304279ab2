"""Dev probe: how much of k_encrypt (DJN, 2^20 x 2048-bit) is the table gather — random r against one r shared by every element
(the same 57 entries for all lanes: cache-resident) and r = 0."""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from bench import synthetic_key
from pailliercryptolib_python_amd import engine
dev = torch.device('cuda', 0)
bits = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
key = synthetic_key(bits, 0x1234567)
pub = engine.PublicKeyHandle(key.n, bits, key.hs, key.randbits, device=dev)
g = torch.Generator(device=dev); g.manual_seed(1)
m = torch.randint(0, 2**31 - 1, (B, pub.n_words), dtype=torch.int64, device=dev, generator=g).to(torch.int32)
m[:, -1] &= 0x0FFFFFFF
r = pub.random_r(B, generator=g)
def tm(f, reps=3):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return round((time.perf_counter() - t0) / reps * 1e3, 2)
row = {"bits": bits, "B": B}
row["random_r_ms"] = tm(lambda: pub.encrypt(m, r))
r1 = r[:1].expand(B, -1).contiguous()
row["shared_r_ms"] = tm(lambda: pub.encrypt(m, r1))
# 64 distinct r per wave tile repeated: every wave gathers 64 different rows, but the working set is 64 x 57 entries (2 MB): L2-resident
r64 = r[:64].repeat(B // 64, 1).contiguous()
row["r_period_64_ms"] = tm(lambda: pub.encrypt(m, r64))
r4k = r[:4096].repeat(B // 4096, 1).contiguous()
row["r_period_4096_ms"] = tm(lambda: pub.encrypt(m, r4k))
print(json.dumps(row))
