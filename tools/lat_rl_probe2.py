"""Dev probe: decrypt latency at N = 16 for whatever library PAI_NATIVE_LIB names (results not checked: debug variants)."""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from bench import synthetic_key
from pailliercryptolib_python_amd import engine
dev = torch.device('cuda', 0)
key = synthetic_key(2048, 0x1234567)
pub = engine.PublicKeyHandle(key.n, 2048, key.hs, key.randbits, device=dev)
priv = engine.PrivateKeyHandle(pub, key.p, key.q)
g = torch.Generator(device=dev); g.manual_seed(1)
N = 16
m = torch.randint(0, 2**31 - 1, (N, pub.n_words), dtype=torch.int64, device=dev, generator=g).to(torch.int32)
m[:, -1] &= 0x0FFFFFFF
ct = pub.encrypt(m, pub.random_r(N, generator=g))
engine.profile_enable(True)
ok = bool(torch.equal(priv.decrypt(ct), m))
ts = []
for _ in range(5):
    priv.decrypt(ct); torch.cuda.synchronize(); ts.append(engine.profile_last())
print(json.dumps({"lib": os.environ.get("PAI_NATIVE_LIB", "default"), "rl": os.environ.get("PAI_LAT_RL", "1"), "ok": ok, "kernel_ms": ts[-1]}))
