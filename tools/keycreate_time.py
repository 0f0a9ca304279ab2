import time, sys, os
sys.path.insert(0, '/root/repo')
import torch
from bench import synthetic_key
from pailliercryptolib_python_amd import engine
key = synthetic_key(2048, 0x1234567)
for w in (8, 10, 11, 12):
    os.environ["PAI_TUNE"] = f"fb_wbits={w}"
    torch.cuda.synchronize(); t = time.time()
    pub = engine.PublicKeyHandle(key.n, 2048, key.hs, key.randbits, device="cuda:0")
    torch.cuda.synchronize(); print("w", w, "pubkey create s", round(time.time() - t, 3), flush=True)
    del pub
