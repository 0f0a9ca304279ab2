"""First-call cost of DJN encryption at the engine level (handle creation, lazy fixed-base table build, steady state):
    python tools/first_call_probe.py   ->  handle 0.07 s, first encrypt of 2^20 0.13 s (0.07 s of it the table; 0.18 s with PAI_DISABLE=fb_chain), then 0.062 s"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
from bench import synthetic_key
from pailliercryptolib_python_amd import engine
dev = torch.device('cuda', 0)
torch.zeros(1, device=dev); torch.cuda.synchronize()
key = synthetic_key(2048, 0x1234567)
t0 = time.perf_counter(); pub = engine.PublicKeyHandle(key.n, 2048, key.hs, key.randbits, device=dev); torch.cuda.synchronize(); print("handle", round(time.perf_counter() - t0, 3))
B = 1 << 20
m = torch.zeros((B, pub.n_words), dtype=torch.int32, device=dev)
gen = torch.Generator(device=dev); gen.manual_seed(1)
r = pub.random_r(B, generator=gen)
out = pub.empty_ct(B); torch.cuda.synchronize()
engine.profile_enable(True)
for i in range(3):
    t0 = time.perf_counter(); pub.encrypt(m, r, out=out); torch.cuda.synchronize(); print("encrypt", i, round(time.perf_counter() - t0, 3), engine.profile_last())
