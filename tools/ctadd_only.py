"""Dev probe: a few ct+ct / ct^-1 calls on a resident 2^20 batch (for rocprofv3 --pmc / --kernel-trace)."""
import sys
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
from bench import synthetic_key
from pailliercryptolib_python_amd import engine
dev = torch.device('cuda', 0)
key = synthetic_key(2048, 0x1234567)
pub = engine.PublicKeyHandle(key.n, 2048, key.hs, key.randbits, device=dev)
B = 1 << 20
g = torch.Generator(device=dev); g.manual_seed(1)
ct = torch.randint(-(2**31), 2**31, (B, pub.ct_words), dtype=torch.int64, device=dev, generator=g).to(torch.int32)
ct[:, -1] &= 0x0FFFFFFF            # < n^2
ct2 = pub.empty_ct(B)
for _ in range(3):
    pub.ct_add(ct, ct2 if False else ct, out=ct2)
torch.cuda.synchronize()
for _ in range(2):
    pub.ct_prod(ct, 1)
torch.cuda.synchronize()
