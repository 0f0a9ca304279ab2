"""Dev probe: where the API-level time goes (host float64 -> encrypt -> decrypt_to_numpy), steady state."""
import os, sys, time, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
from bench import synthetic_key
from pailliercryptolib_python_amd import PaillierPrivateKey, PaillierPublicKey, engine
from pailliercryptolib_python_amd import fixedpoint as _fp
from pailliercryptolib_python_amd.bindings import ipclPublicKey
key = synthetic_key(2048, 0x1234567)
pk = PaillierPublicKey(ipclPublicKey(key.n, 2048, True, hs=key.hs, randbits=key.randbits))
sk = PaillierPrivateKey(pk, key.p, key.q)
B = 1 << 20
x = np.random.default_rng(1).uniform(-1000, 1000, B)
def sync(): torch.cuda.synchronize()
for rep in range(3):
    t = {}
    sync(); t0 = time.perf_counter()
    en = pk.encrypt(x); sync(); t['encrypt_total'] = time.perf_counter() - t0
    h = pk.pubkey.handle
    t0 = time.perf_counter(); xs = torch.from_numpy(_fp.checked_float64(x)).to(h.device); sync(); t['  h2d+check'] = time.perf_counter() - t0
    t0 = time.perf_counter(); m, e = h.fp_encode_f64(xs); sync(); t['  encode'] = time.perf_counter() - t0
    t0 = time.perf_counter(); ex = e.cpu().numpy(); t['  expo d2h'] = time.perf_counter() - t0
    t0 = time.perf_counter(); r = pk.pubkey._draw_r(B); sync(); t['  draw_r'] = time.perf_counter() - t0
    t0 = time.perf_counter(); ct = h.encrypt(m, r); sync(); t['  pai_encrypt'] = time.perf_counter() - t0
    sync(); t0 = time.perf_counter()
    back = sk.decrypt_to_numpy(en); t['decrypt_total'] = time.perf_counter() - t0
    t0 = time.perf_counter(); pt = sk.prikey.handle.decrypt(en.words); sync(); t['  pai_decrypt'] = time.perf_counter() - t0
    t0 = time.perf_counter(); mant, flag = h.fp_decode_i64(pt); sync(); t['  decode'] = time.perf_counter() - t0
    t0 = time.perf_counter(); bad = bool(flag.any()); t['  flag.any'] = time.perf_counter() - t0
    t0 = time.perf_counter(); mh = mant.cpu().numpy(); t['  mant d2h'] = time.perf_counter() - t0
    t0 = time.perf_counter(); y = np.ldexp(mh.astype(np.float64), -np.asarray(en._expo, dtype=np.int64).astype(np.int32)); t['  ldexp'] = time.perf_counter() - t0
    assert np.array_equal(back, x)
    print(json.dumps({k: round(v * 1e3, 2) for k, v in t.items()}))
