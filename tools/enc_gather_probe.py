"""Probe: is the digit-engine encryption kernel limited by the fixed-base table gather?  Times pai_encrypt with
random r (every lane reads a different table entry) and with r = 0 (every lane reads entry 0 of each window)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import synthetic_key
from pailliercryptolib_python_amd import engine, fixedpoint
key = synthetic_key(2048, 0x1234567)
pub = engine.PublicKeyHandle(key.n, 2048, key.hs, key.randbits, device="cuda:0")
B = 1 << 18
x = np.random.default_rng(1).uniform(-1000, 1000, B)
res, _ = fixedpoint.encode_float64_array(x, key.n, pub.n_words)
m = engine.to_device_words(res, pub.device)
for name, r in (("random r", pub.random_r(B)), ("r = 0", torch.zeros((B, pub.r_words), dtype=torch.int32, device=pub.device)),
                ("r = same", pub.random_r(1).expand(B, -1).contiguous())):
    ct = pub.encrypt(m, r); torch.cuda.synchronize()
    t = time.perf_counter(); ct = pub.encrypt(m, r); torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(f"{name:10s}: {dt*1e3:8.2f} ms for {B} elements  ({os.environ.get('PAI_ENABLE_PADIC_ENC','0')=}, {os.environ.get('PAI_FB_WBITS','auto')=})", flush=True)
