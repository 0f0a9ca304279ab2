import sys, time, json
sys.path.insert(0, '.')
import numpy as np, torch
from bench import synthetic_key
from pailliercryptolib_python_amd import engine, fixedpoint
dev = torch.device('cuda', 0)
key = synthetic_key(2048, None)
pub = engine.PublicKeyHandle(key.n, 2048, None, 0, device=dev)
priv = engine.PrivateKeyHandle(pub, key.p, key.q)
B = 65536
x = np.random.default_rng(7).uniform(-1000.0, 1000.0, B)
res, _ = fixedpoint.encode_float64_array(x, key.n, pub.n_words)
m = engine.to_device_words(res, dev)
r = torch.randint(-(2**31), 2**31, (B, pub.n_words), dtype=torch.int64, device=dev).to(torch.int32)
r[:, -1] &= 0x3FFFFFFF
r[:, 0] |= 1
ct = pub.encrypt(m, r); torch.cuda.synchronize()
t0 = time.perf_counter(); ct = pub.encrypt(m, r); torch.cuda.synchronize(); t = time.perf_counter() - t0
out = priv.decrypt(ct)
idx = [0, 5, B - 1]
r_h = engine.words_to_ints(engine.to_host_words(r[idx])); m_h = engine.words_to_ints(res[idx])
ok = engine.words_to_ints(engine.to_host_words(ct[idx])) == [(1 + mm * key.n) * pow(rr, key.n, key.nsq) % key.nsq for mm, rr in zip(m_h, r_h)]
print(json.dumps({"standard_scheme_encrypt_ms_per_65536": round(t * 1e3, 1), "roundtrip": bool(torch.equal(out, m)), "bits_ok": ok}))
