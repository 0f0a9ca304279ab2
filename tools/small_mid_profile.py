"""Workload for tools/profile_cmd.sh: the small-batch and mid-size kernels at 2048-bit keys — decrypt / DJN encrypt / ct * pt (53-bit)
of 16 and of 8 192 elements, ten calls each (rocprofv3 then lists k_dec_a_pp, k_ctmul_pp, k_encrypt_tree, k_pair_ctmul, k_pair_fixed_base)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from bench import synthetic_key
from pailliercryptolib_python_amd import engine
dev = torch.device('cuda', 0)
key = synthetic_key(2048, 0x1234567)
pub = engine.PublicKeyHandle(key.n, 2048, key.hs, key.randbits, device=dev)
priv = engine.PrivateKeyHandle(pub, key.p, key.q)
g = torch.Generator(device=dev); g.manual_seed(1)
for N in (16, 8192):
    m = torch.randint(0, 2**31 - 1, (N, pub.n_words), dtype=torch.int64, device=dev, generator=g).to(torch.int32)
    m[:, -1] &= 0x0FFFFFFF
    r = pub.random_r(N, generator=g)
    e = torch.randint(-2**31, 2**31 - 1, (N, 2), dtype=torch.int64, device=dev, generator=g).to(torch.int32)
    e[:, 1] &= (1 << 21) - 1
    e[:, 1] |= 1 << 20
    for _ in range(10):
        ct = pub.encrypt(m, r)
        back = priv.decrypt(ct)
        pw = pub.ct_mul(ct, e, 53)
    torch.cuda.synchronize()
    assert torch.equal(back, m)
print("ok")
