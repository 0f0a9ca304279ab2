"""Dev probe: ct_invert / subtraction latency at small and large batches."""
import os, sys, time, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
from bench import synthetic_key
from pailliercryptolib_python_amd import engine
dev = torch.device('cuda', 0)
for bits in (2048, 4096):
    key = synthetic_key(bits, 0x1234567)
    pub = engine.PublicKeyHandle(key.n, bits, key.hs, key.randbits, device=dev)
    g = torch.Generator(device=dev); g.manual_seed(1)
    row = {"bits": bits}
    for N in (16, 1 << 16):
        ct = torch.randint(-(2**31), 2**31, (N, pub.ct_words), dtype=torch.int64, device=dev, generator=g).to(torch.int32)
        ct[:, -1] &= 0x0FFFFFFF
        ct[:, 0] |= 1
        def f(): pub.ct_invert(ct)
        f(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): f()
        torch.cuda.synchronize()
        row[f"inv_{N}_ms"] = round((time.perf_counter() - t0) / 5 * 1e3, 3)
        inv = pub.ct_invert(ct)
        one = engine.words_to_ints(engine.to_host_words(pub.ct_add(ct[:2].contiguous(), inv[:2].contiguous())))
        assert one == [1, 1], one
    print(json.dumps(row))
