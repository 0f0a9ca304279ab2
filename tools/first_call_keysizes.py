"""Dev probe: first DJN encryption (fixed-base table build included) vs steady state per key size; PAI_DISABLE=fb_chain gives
the table kernels' own squaring chains back."""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from bench import synthetic_key
from pailliercryptolib_python_amd import engine
dev = torch.device('cuda', 0)
torch.zeros(1, device=dev); torch.cuda.synchronize()
for bits in [int(a) for a in sys.argv[1:]] or [1024, 2048, 3072, 4096]:
    key = synthetic_key(bits, 0x1234567)
    pub = engine.PublicKeyHandle(key.n, bits, key.hs, key.randbits, device=dev)
    B = 1 << 16
    m = torch.zeros((B, pub.n_words), dtype=torch.int32, device=dev)
    gen = torch.Generator(device=dev); gen.manual_seed(1)
    r = pub.random_r(B, generator=gen)
    out = pub.empty_ct(B); torch.cuda.synchronize()
    ts = []
    for i in range(3):
        t0 = time.perf_counter(); pub.encrypt(m, r, out=out); torch.cuda.synchronize(); ts.append(round(time.perf_counter() - t0, 4))
    print(json.dumps({"bits": bits, "batch": B, "first_s": ts[0], "steady_s": ts[2], "chain": "fb_chain" not in os.environ.get("PAI_DISABLE", "")}), flush=True)
    del pub
