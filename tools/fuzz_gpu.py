"""Randomised differential run of the C ABI against CPython integers (dev tool; the committed tests are the fixed cases).
Random batch sizes, operand patterns with long carry chains (all-ones runs, values next to 0 / M), every key size.
    python tools/fuzz_gpu.py [seconds] [seed]        (tests/test_gpu_fuzz.py runs a bounded, seeded pass in the driver's suite)"""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from oracle import paillier_oracle as orc
from pailliercryptolib_python_amd import _native
from tests._util import DevArray, host_ptr, ints_to_limbs, limbs_to_ints
from tests.test_gpu_paillier_abi import NativeKey, bench_key, seeded_key

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
SEED = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
rng = np.random.default_rng(SEED)
KEY_BITS = (1024, 1536, 2048, 2560, 3072, 3328, 3584, 4096)      # 3328 / 3584: either side of the one- / two-limb chain layouts of the small-batch pipeline
def key_for(b):
    if b == 2048: return bench_key()
    if b in (1024, 3072, 4096): return seeded_key(b)                  # the committed fixture primes
    p, q = _native.keygen(b, True, seed=b)                             # the other sizes: the native generator, seeded
    return orc.make_key(p, q, djn_x=(1 << 70) + 12345, bits=b)
keys = {b: NativeKey(key_for(b)) for b in KEY_BITS}

def pattern(M, n):
    out = []
    bits = M.bit_length()
    for _ in range(n):
        k = rng.integers(0, 8)
        if k == 0: v = int(rng.integers(0, 3))
        elif k == 1: v = M - 1 - int(rng.integers(0, 3))
        elif k == 2: v = (1 << int(rng.integers(1, bits))) - 1                      # run of ones
        elif k == 3: v = ((1 << bits) - 1) ^ ((1 << int(rng.integers(1, bits))) - 1)  # ones on top
        elif k == 4: v = 1 << int(rng.integers(0, bits))
        else: v = int.from_bytes(rng.bytes(bits // 8 + 8), "little")
        out.append(v % M)
    return out

t0 = time.time(); rounds = 0; checks = 0
while time.time() - t0 < budget:
    bits = int(rng.choice(KEY_BITS))
    nk = keys[bits]; key = nk.key; M = key.nsq; lib = nk.lib
    N = int(rng.integers(1, 700))
    a, b = pattern(M, N), pattern(M, N)
    da, db = DevArray(ints_to_limbs(a, nk.cw)), DevArray(ints_to_limbs(b, nk.cw))
    out = DevArray(shape=(N, nk.cw))
    bc = int(rng.integers(0, 2))
    if rng.integers(0, 2): os.environ["PAI_DISABLE"] = "add_msb"         # two Montgomery products instead of the most-significant-limb-first one
    if rng.integers(0, 2): os.environ["PAI_LAT_ADD_MAX"] = "0"           # lane groups (wave tiles) at any batch size
    _native.check(lib.pai_ct_add(nk.pk, da.ptr, db.ptr, bc, N, out.ptr, None))
    os.environ.pop("PAI_DISABLE", None)
    os.environ.pop("PAI_LAT_ADD_MAX", None)
    assert limbs_to_ints(out.get()) == [x * (b[0] if bc else y) % M for x, y in zip(a, b)], ("ct_add", bits, N, bc)
    # lazy Montgomery domain: the single product, and the aligned addition on a random common tag
    rb = C.c_int(0)
    _native.check(lib.pai_pubkey_mont_bits(nk.pk, C.byref(rb)))
    Rm = pow(2, rb.value, M); Ri = pow(Rm, -1, M)
    _native.check(lib.pai_ct_mont_mul(nk.pk, da.ptr, db.ptr, bc, N, out.ptr, None))
    assert limbs_to_ints(out.get()) == [x * (b[0] if bc else y) * Ri % M for x, y in zip(a, b)], ("ct_mont_mul", bits, N, bc)
    kt = int(rng.integers(-3, 4))
    rk = pow(Rm, kt, M) if kt >= 0 else pow(Ri, -kt, M)
    r2k = pow(Rm, 2 - kt, M) if 2 - kt >= 0 else pow(Ri, kt - 2, M)
    dlt = rng.integers(-4, 5, N).astype(np.int32)
    dak, dbk = DevArray(ints_to_limbs([x * rk % M for x in a], nk.cw)), DevArray(ints_to_limbs([y * rk % M for y in b], nk.cw))
    ddlt, dent = DevArray(dlt), DevArray(ints_to_limbs([r2k], nk.cw))
    _native.check(lib.pai_ct_add_aligned_dom(nk.pk, dak.ptr, dbk.ptr, 0, ddlt.ptr, N, out.ptr, dent.ptr, None))
    assert limbs_to_ints(out.get()) == [(x * pow(y, 1 << int(t), M) if t > 0 else pow(x, 1 << int(-t), M) * y) * rk % M
                                        for x, y, t in zip(a, b, dlt)], ("add_aligned_dom", bits, N, kt)
    groups = int(rng.choice([g for g in (1, 2, 3, 5, 7, 16) if N % g == 0] or [1])) if N > 1 else 1
    if N % groups == 0:
        og = DevArray(shape=(groups, nk.cw))
        _native.check(lib.pai_ct_prod(nk.pk, da.ptr, N, groups, og.ptr, None))
        want = []
        for g in range(groups):
            p = 1
            for l in range(N // groups): p = p * a[l * groups + g] % M
            want.append(p)
        assert limbs_to_ints(og.get()) == want, ("ct_prod", bits, N, groups)
    delta = rng.integers(-3, 9, N).astype(np.int32)
    dd = DevArray(delta); dc = DevArray(ints_to_limbs(a, nk.cw))
    if rng.integers(0, 2):
        _native.check(lib.pai_ct_pow2(nk.pk, dc.ptr, dd.ptr, 0, N, None))
    else:
        _native.check(lib.pai_ct_pow2_hint(nk.pk, dc.ptr, dd.ptr, 0, N, int(max(0, delta.max())), None))
    assert limbs_to_ints(dc.get()) == [pow(x, 1 << int(d), M) if d > 0 else x for x, d in zip(a, delta)], ("pow2", bits, N)
    units = [x if (x % key.p and x % key.q) else 1 for x in a]
    du = DevArray(ints_to_limbs(units, nk.cw))
    _native.check(lib.pai_ct_invert(nk.pk, du.ptr, N, out.ptr, None))
    got = limbs_to_ints(out.get())
    for i in range(0, N, max(1, N // 40)): assert got[i] == pow(units[i], -1, M), ("invert", bits, N, i)
    # decrypt on both paths + ct*pt on both paths, small N
    n2 = int(rng.integers(1, 40))
    m = pattern(key.n, n2)
    cts = [orc.encrypt(key, x, int(rng.integers(1, 1 << 62))) for x in m]
    dct = DevArray(ints_to_limbs(cts, nk.cw)); om = DevArray(shape=(n2, nk.nw))
    ebits = int(rng.choice([9, 31, 53, 64, 200]))
    e = [int.from_bytes(rng.bytes(ebits // 8 + 1), "little") % (1 << ebits) for _ in range(n2)]
    ew = (ebits + 31) // 32
    de = DevArray(ints_to_limbs(e, ew)); oc = DevArray(shape=(n2, nk.cw))
    rr = pattern(1 << key.randbits, n2)
    want_enc = [orc.encrypt(key, x, y) for x, y in zip(m, rr)]
    dmm, drr = DevArray(ints_to_limbs(m, nk.nw)), DevArray(ints_to_limbs(rr, nk.rw))
    oe = DevArray(shape=(n2, nk.cw))
    for sw in ("0", "100000"):
        os.environ["PAI_LATENCY_MAX"] = sw
        os.environ["PAI_TUNE"] = (f"lat_pp={int(rng.choice([0, 100000]))},lat_rl={int(rng.choice([0, 100000]))},"   # every small-batch stage A
                                 f"lat_mul_pp={int(rng.choice([0, 100000]))},lat_mul_rl={int(rng.choice([0, 100000]))}"   # ... and ct * pt kernel
                                 + (",dec_mid_min=0,dec_mid_max=1000000" if rng.integers(0, 3) == 0 else "")   # lane-group digit-pair stage A
                                 + (",ctmul_mid_min=0,ctmul_mid_max=1000000" if bits <= 2048 and rng.integers(0, 3) == 0 else "")   # ... and ct * pt
                                 + (",enc_mid_min=0,enc_mid_max=1000000" if bits <= 2048 and rng.integers(0, 3) == 0 else ""))   # ... and DJN encryption
        _native.check(lib.pai_encrypt(nk.pk, dmm.ptr, drr.ptr, n2, oe.ptr, None))
        assert limbs_to_ints(oe.get()) == want_enc, ("encrypt", bits, n2, sw)
        _native.check(lib.pai_decrypt(nk.sk, dct.ptr, n2, om.ptr, None))
        assert limbs_to_ints(om.get()) == m, ("decrypt", bits, n2, sw)
        _native.check(lib.pai_ct_mul(nk.pk, dct.ptr, de.ptr, ew, ebits, 0, n2, oc.ptr, None))
        assert limbs_to_ints(oc.get()) == [pow(c, x, M) for c, x in zip(cts, e)], ("ct_mul", bits, n2, ebits, sw)
    os.environ.pop("PAI_LATENCY_MAX", None)
    os.environ.pop("PAI_TUNE", None)
    # n-ary sum: random operand count, domain tags, result tag and exponent raises
    kk = int(rng.integers(2, 17)); Nn = int(rng.integers(1, 200))
    t0_, t_ = int(rng.integers(-3, 4)), int(rng.integers(-1, 2))      # (|1 + dom_out - c| must stay within the table of R^m: 48)
    dom_out = int(rng.integers(-6, 7))
    def rp(m_): return pow(Rm, m_, M) if m_ >= 0 else pow(Ri, -m_, M)
    vals = [pattern(M, Nn) for _ in range(kk)]
    rz = [None if rng.integers(0, 3) == 0 else rng.integers(0, 4, Nn).astype(np.int32) for _ in range(kk)]
    if rz[0] is not None: t0_ = t_
    ops = [DevArray(ints_to_limbs([x * rp(t0_ if j == 0 else t_) % M for x in v], nk.cw)) for j, v in enumerate(vals)]
    rzd = [None if r_ is None else DevArray(r_) for r_ in rz]
    ptrs = (C.c_void_p * kk)(*[o.ptr.value for o in ops])
    rzp = (C.c_void_p * kk)(*[None if r_ is None else r_.ptr.value for r_ in rzd])
    on = DevArray(shape=(Nn, nk.cw))
    _native.check(lib.pai_ct_addn(nk.pk, ptrs, rzp, kk, t0_, t_, dom_out, Nn, on.ptr, None))
    wantn = []
    for i in range(Nn):
        acc = 1
        for j in range(kk): acc = acc * pow(vals[j][i], 1 << (0 if rz[j] is None else int(rz[j][i])), M) % M
        wantn.append(acc * rp(dom_out) % M)
    assert limbs_to_ints(on.get()) == wantn, ("ct_addn", bits, kk, Nn, t0_, t_, dom_out)
    # multi-exponentiation (digit engine up to 2048-bit keys, lane groups above) with forced chunking and random table
    # widths, and pow2 on the digit engine
    if True:
        R, K, Mc = int(rng.integers(1, 4)), int(rng.integers(1, 24)), int(rng.integers(1, 6))
        base = [x if (x % key.p and x % key.q) else 5 for x in pattern(M, R * K)]
        inv = [pow(x, -1, M) for x in base]
        eb = int(rng.choice([1, 7, 53, 64, 75, 100]))
        ew2 = (eb + 31) // 32
        ee = [[[int.from_bytes(rng.bytes(eb // 8 + 1), "little") % (1 << eb) for _ in range(Mc)] for _ in range(K)] for _ in range(R)]
        sg = rng.integers(0, 2, size=(K, Mc)).astype(np.uint8)
        e_l = np.zeros((R, K, Mc, ew2), dtype=np.uint32)
        for r_ in range(R):
            for l_ in range(K):
                for j_ in range(Mc):
                    for w_ in range(ew2): e_l[r_, l_, j_, w_] = (ee[r_][l_][j_] >> (32 * w_)) & 0xFFFFFFFF
        os.environ["PAI_TUNE"] = f"mexp_lanes={int(rng.integers(1, 50))},mexp_wbits={int(rng.integers(1, 8))}"
        dcb, dib, deb, dsb = DevArray(ints_to_limbs(base, nk.cw)), DevArray(ints_to_limbs(inv, nk.cw)), DevArray(e_l), DevArray(sg)
        ob = DevArray(shape=(R * Mc, nk.cw))
        _native.check(lib.pai_ct_multiexp(nk.pk, dcb.ptr, dib.ptr, R, K, Mc, deb.ptr, ew2, eb, dsb.ptr, ob.ptr, None))
        wantb = []
        for r_ in range(R):
            for j_ in range(Mc):
                acc = 1
                for l_ in range(K): acc = acc * pow(inv[r_ * K + l_] if sg[l_, j_] else base[r_ * K + l_], ee[r_][l_][j_], M) % M
                wantb.append(acc)
        assert limbs_to_ints(ob.get()) == wantb, ("multiexp", bits, R, K, Mc, eb)
    if bits <= 2048:
        os.environ["PAI_POW2_DIGIT_MIN"] = "1"
        dl2 = rng.integers(-3, 63, N).astype(np.int32)
        dd2 = DevArray(dl2); dc2 = DevArray(ints_to_limbs(a, nk.cw))
        _native.check(lib.pai_ct_pow2(nk.pk, dc2.ptr, dd2.ptr, 0, N, None))
        got2 = limbs_to_ints(dc2.get())
        for i in range(0, N, max(1, N // 30)):
            assert got2[i] == (pow(a[i], 1 << int(dl2[i]), M) if dl2[i] > 0 else a[i]), ("pow2_digit", bits, N, i)
        os.environ.pop("PAI_POW2_DIGIT_MIN", None)
    rounds += 1; checks += 13
print(json.dumps({"rounds": rounds, "checks": checks, "seconds": round(time.time() - t0, 1), "failures": 0, "seed": SEED}))
