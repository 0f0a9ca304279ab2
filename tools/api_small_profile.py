"""Dev probe: where the microseconds of a small public-API call go — the reference's own benchmark rows (bench/bench_ipcl_python.py:22-78)
at 16 elements, each call followed by a device synchronisation as bench.py: reference_bench times them; then cProfile over 400 calls each.
python tools/api_small_profile.py [rows=all|add_ctct,...] [elements=16]"""
import cProfile, pstats, io, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
from bench import synthetic_key
from pailliercryptolib_python_amd import PaillierPublicKey, PaillierPrivateKey
from pailliercryptolib_python_amd.bindings import ipclPublicKey
key = synthetic_key(2048, 0x1234567)
pk = PaillierPublicKey(ipclPublicKey(key.n, 2048, True, hs=key.hs, randbits=key.randbits, device=torch.device("cuda", 0)))
sk = PaillierPrivateKey(pk, key.p, key.q)
want = sys.argv[1].split(",") if len(sys.argv) > 1 and sys.argv[1] != "all" else None
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 16
ar = np.arange(nb)
x_enc, x_dec = (ar + 11) * 1234.5678, (ar + 1) * 1234.5678
x, y = (ar + 11) * 5111.2834, (32768 - ar) * 1.3872
ct_dec, ct_x, ct_y = pk.encrypt(x_dec), pk.encrypt(x), pk.encrypt(y)
ct_xx = ct_x * x
rows = {
    "encrypt": lambda: pk.encrypt(x_enc),
    "decrypt": lambda: sk.decrypt(ct_dec),
    "add_ctct": lambda: (ct_x + ct_y).words,
    "add_ctct_lazy": lambda: ct_x + ct_y,
    "add_ctpt": lambda: (ct_xx + y).words,
    "mul_ctpt": lambda: ct_x * y,
}
def timeit(name, f, reps=400):
    for _ in range(3): f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        f(); torch.cuda.synchronize()
    print(f"{name:16s} {(time.perf_counter() - t0) / reps * 1e6:8.1f} us per call (synchronised)")
def issue_rate(name, f, reps=400):
    """unsynchronised: per-call cost when calls are issued back to back = max(host work per call, device work per call)"""
    for _ in range(3): f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"{name:16s} host issue {t_issue / reps * 1e6:8.1f} us per call, with the final drain {t_all / reps * 1e6:8.1f} us")
def kernels(name, f):
    from pailliercryptolib_python_amd import engine
    engine.profile_enable(True)
    f(); torch.cuda.synchronize()
    print(f"{name:16s} kernels (last C call): {engine.profile_last()}")
    engine.profile_enable(False)
for name, f in rows.items():
    if want is None or name in want: timeit(name, f)
for name, f in rows.items():
    if want is None or name in want: issue_rate(name, f)
for name, f in rows.items():
    if want is None or name in want: kernels(name, f)
for name, f in rows.items():
    if want is not None and name not in want: continue
    if want is None and name in ("add_ctct_lazy",): continue
    pr = cProfile.Profile(); pr.enable()
    for _ in range(400):
        f(); torch.cuda.synchronize()
    pr.disable()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22)
    print("=====", name); print("\n".join(l[:150] for l in s.getvalue().splitlines()[:40]))
