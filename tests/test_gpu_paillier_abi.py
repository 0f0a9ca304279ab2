"""GPU parity of the Paillier hot path through the C ABI against the Python-int oracle
(oracle/paillier_oracle.py).  Bit-exact ciphertexts with explicit randomness; bit-exact plaintexts."""
import ctypes as C
import json
from pathlib import Path

import numpy as np
import pytest

from oracle import paillier_oracle as orc
from pailliercryptolib_python_amd import _native
from tests._util import DevArray, djn_encrypt_many, djn_obfuscate_many, host_ptr, ints_to_limbs, limbs_to_ints, pow_many, rand_below, tune
from tests._util import disable as knob_disable

pytestmark = pytest.mark.gpu


class NativeKey:
    def __init__(self, key: orc.OracleKey):
        self.lib = _native.load()
        self.key = key
        self.nw = (key.bits + 31) // 32
        self.cw = 2 * self.nw
        pk = C.c_void_p()
        n_l = ints_to_limbs([key.n], self.nw)
        if key.hs is not None:
            hs_l = ints_to_limbs([key.hs], self.cw)
            _native.check(self.lib.pai_pubkey_create(host_ptr(n_l), self.nw, key.bits, host_ptr(hs_l), self.cw,
                                                     key.randbits, 0, C.byref(pk)))
            self.rw = (key.randbits + 31) // 32
        else:
            _native.check(self.lib.pai_pubkey_create(host_ptr(n_l), self.nw, key.bits, None, 0, 0, 0, C.byref(pk)))
            self.rw = self.nw
        self.pk = pk
        sk = C.c_void_p()
        pw = (key.q.bit_length() + 31) // 32
        # deliberately pass the primes in the "wrong" order (bench passes P > Q)
        _native.check(self.lib.pai_privkey_create(pk, host_ptr(ints_to_limbs([key.q], pw)), pw,
                                                  host_ptr(ints_to_limbs([key.p], pw)), pw, C.byref(sk)))
        self.sk = sk

    def __del__(self):
        try:
            self.lib.pai_privkey_destroy(self.sk)
            self.lib.pai_pubkey_destroy(self.pk)
        except Exception:
            pass


def bench_key(djn=True):
    return orc.make_key(orc.BENCH_P, orc.BENCH_Q, djn_x=0x1234567 if djn else None, bits=2048)


def seeded_key(bits, djn=True):
    fx = json.loads((Path(__file__).parent / "golden" / "fixture_keys.json").read_text())[str(bits)]
    return orc.make_key(int(fx["p"], 16), int(fx["q"], 16), djn_x=(1 << 70) + 12345 if djn else None, bits=bits)


@pytest.fixture(scope="module")
def k2048():
    return NativeKey(bench_key())


def plaintexts(key, N, seed):
    rng = np.random.default_rng(seed)
    m = rand_below(rng, key.n, N)
    m[0], m[1], m[2] = 0, 1, key.n - 1
    return m


@pytest.mark.parametrize("lat_add", ["0", "4096"])
def test_raw_encrypt_and_decrypt_2048(k2048, lat_add, monkeypatch):
    monkeypatch.setenv("PAI_LAT_ADD_MAX", lat_add)       # small raw encryptions: one product on the latency geometry, or the digit engine
    key, N = k2048.key, 300
    m = plaintexts(key, N, 1)
    dm = DevArray(ints_to_limbs(m, k2048.nw))
    ct = DevArray(shape=(N, k2048.cw))
    _native.check(k2048.lib.pai_raw_encrypt(k2048.pk, dm.ptr, N, ct.ptr, None))
    got = limbs_to_ints(ct.get())
    assert got == [orc.raw_encrypt(x, key.n) for x in m]
    out = DevArray(shape=(N, k2048.nw))
    _native.check(k2048.lib.pai_decrypt(k2048.sk, ct.ptr, N, out.ptr, None))
    assert limbs_to_ints(out.get()) == m


def test_djn_encrypt_bits_and_roundtrip_2048(k2048):
    key, N = k2048.key, 257
    m = plaintexts(key, N, 2)
    r_l = orc.synth_r_limbs(4002, N, key.randbits)
    r_l[0] = 0                      # r = 0 -> obfuscator 1
    r_l[1] = 0xFFFFFFFF             # r = 2^randbits - 1
    r = limbs_to_ints(r_l)
    dm, dr = DevArray(ints_to_limbs(m, k2048.nw)), DevArray(r_l)
    ct = DevArray(shape=(N, k2048.cw))
    _native.check(k2048.lib.pai_encrypt(k2048.pk, dm.ptr, dr.ptr, N, ct.ptr, None))
    got = limbs_to_ints(ct.get())
    want = [orc.encrypt(key, x, rr) for x, rr in zip(m, r)]
    assert got == want
    out = DevArray(shape=(N, k2048.nw))
    _native.check(k2048.lib.pai_decrypt(k2048.sk, ct.ptr, N, out.ptr, None))
    assert limbs_to_ints(out.get()) == m
    # decrypt agrees with the non-CRT definition too
    assert orc.decrypt_lambda(key, want[5]) == m[5]
    # apply_obfuscator on existing ciphertexts
    r2_l = orc.synth_r_limbs(4003, N, key.randbits)
    dr2 = DevArray(r2_l)
    _native.check(k2048.lib.pai_obfuscate(k2048.pk, ct.ptr, dr2.ptr, N, None))
    assert limbs_to_ints(ct.get()) == [orc.apply_obfuscator(key, c, rr) for c, rr in zip(want, limbs_to_ints(r2_l))]


@pytest.mark.parametrize("wbits", ["5", "12", "14", "16"])
def test_djn_encrypt_every_table_geometry_2048(wbits, monkeypatch):
    """The digit-form fixed-base table is built directly (<= 12 bits, odd widths included) or in two levels
    (even widths above 12; the default picks 18 bits on a 288 GB device): the ciphertext bits must not
    depend on the geometry."""
    tune(monkeypatch, "fb_digit_wbits", wbits)
    nk = NativeKey(bench_key())
    key, N = nk.key, 130
    m = plaintexts(key, N, 21)
    r_l = orc.synth_r_limbs(4021, N, key.randbits)
    r_l[0] = 0
    r_l[1] = 0xFFFFFFFF
    dm, dr = DevArray(ints_to_limbs(m, nk.nw)), DevArray(r_l)
    ct = DevArray(shape=(N, nk.cw))
    _native.check(nk.lib.pai_encrypt(nk.pk, dm.ptr, dr.ptr, N, ct.ptr, None))
    assert limbs_to_ints(ct.get()) == [orc.encrypt(key, x, rr) for x, rr in zip(m, limbs_to_ints(r_l))]


def test_ct_add_mul_pow2_2048(k2048):
    key, N = k2048.key, 200
    rng = np.random.default_rng(7)
    a = rand_below(rng, key.nsq, N)
    b = rand_below(rng, key.nsq, N)
    da, db = DevArray(ints_to_limbs(a, k2048.cw)), DevArray(ints_to_limbs(b, k2048.cw))
    out = DevArray(shape=(N, k2048.cw))
    _native.check(k2048.lib.pai_ct_add(k2048.pk, da.ptr, db.ptr, 0, N, out.ptr, None))
    assert limbs_to_ints(out.get()) == [orc.ct_add(x, y, key.nsq) for x, y in zip(a, b)]
    _native.check(k2048.lib.pai_ct_add(k2048.pk, da.ptr, db.ptr, 1, N, out.ptr, None))
    assert limbs_to_ints(out.get()) == [orc.ct_add(x, b[0], key.nsq) for x in a]
    es = [int(x) for x in rng.integers(0, 1 << 53, size=N)]
    es[0], es[1] = 0, 1
    de = DevArray(ints_to_limbs(es, 2))
    _native.check(k2048.lib.pai_ct_mul(k2048.pk, da.ptr, de.ptr, 2, 53, 0, N, out.ptr, None))
    assert limbs_to_ints(out.get()) == [orc.ct_mul(x, e, key.nsq) for x, e in zip(a, es)]
    delta = rng.integers(-5, 60, size=N).astype(np.int32)
    delta[:4] = [0, 1, -3, 59]
    dd = DevArray(delta)
    _native.check(k2048.lib.pai_ct_pow2(k2048.pk, da.ptr, dd.ptr, 0, N, None))
    assert limbs_to_ints(da.get()) == [orc.ct_mul(x, 2 ** int(d), key.nsq) if d > 0 else x for x, d in zip(a, delta)]


@pytest.mark.parametrize("bits", [1024, 2048])
def test_ct_pow2_through_the_digit_engine(bits, monkeypatch):
    """Large batches with shifts of 8 and more run ct^(2^delta) as ct * pt with a one-bit exponent on the base-n digit
    engine (PAI_POW2_DIGIT_MIN lowers the batch threshold for the test): per-element shifts incl. <= 0 and the maximum
    62, a broadcast shift, and shift sets that stay on the lane-group kernel (all < 8, none positive, one of 63)."""
    monkeypatch.setenv("PAI_POW2_DIGIT_MIN", "1")
    nk = NativeKey(bench_key() if bits == 2048 else seeded_key(bits))
    key, N = nk.key, 300
    rng = np.random.default_rng(bits)
    a = rand_below(rng, key.nsq, N)
    a[0], a[1] = 1, key.nsq - 1
    cases = []
    d = rng.integers(-5, 62, size=N).astype(np.int32)
    d[:5] = [0, 1, -3, 62, 8]
    cases.append((d, 0))
    cases.append((np.array([52], dtype=np.int32), 1))
    cases.append((rng.integers(-2, 8, size=N).astype(np.int32), 0))
    cases.append((np.zeros(N, dtype=np.int32) - 1, 0))
    d63 = rng.integers(0, 20, size=N).astype(np.int32)
    d63[7] = 63
    cases.append((d63, 0))
    for delta, bc in cases:
        da, dd = DevArray(ints_to_limbs(a, nk.cw)), DevArray(delta)
        _native.check(nk.lib.pai_ct_pow2(nk.pk, da.ptr, dd.ptr, bc, N, None))
        dl = [int(delta[0])] * N if bc else [int(v) for v in delta]
        assert limbs_to_ints(da.get()) == [pow(x, 1 << v, key.nsq) if v > 0 else x for x, v in zip(a, dl)], (bits, bc, int(delta.max()))


@pytest.mark.parametrize("ebits", [1, 2, 24, 25, 80, 81, 240, 241, 2048, 4096])
def test_ct_mul_every_window_width_and_exponent_shape(k2048, ebits):
    """ct^e on the base-n digit engine: the window width follows ebits_max (2/3/4/5 bits), exponents of every
    size up to a full ciphertext width, zero windows, e = 0, per-element and broadcast exponents, in place."""
    key, N = k2048.key, 70
    rng = np.random.default_rng(900 + ebits)
    a = rand_below(rng, key.nsq, N)
    a[0], a[1] = 1, key.nsq - 1
    es = [int.from_bytes(rng.bytes(ebits // 8 + 1), "little") % (1 << ebits) for _ in range(N)]
    es[0], es[1], es[2] = 0, (1 << ebits) - 1, 1 << (ebits - 1)
    if ebits > 8:
        es[3] = 1 << (ebits - 1) | 1                     # one long run of zero windows
    ew = (ebits + 31) // 32
    da, de = DevArray(ints_to_limbs(a, k2048.cw)), DevArray(ints_to_limbs(es, ew))
    out = DevArray(shape=(N, k2048.cw))
    _native.check(k2048.lib.pai_ct_mul(k2048.pk, da.ptr, de.ptr, ew, ebits, 0, N, out.ptr, None))
    assert limbs_to_ints(out.get()) == pow_many(a, es, key.nsq)
    db = DevArray(ints_to_limbs([es[3 if ebits > 8 else 1]], ew))
    _native.check(k2048.lib.pai_ct_mul(k2048.pk, da.ptr, db.ptr, ew, ebits, 1, N, da.ptr, None))     # broadcast, in place
    assert limbs_to_ints(da.get()) == pow_many(a, es[3 if ebits > 8 else 1], key.nsq)


@pytest.mark.parametrize("bits", [1024, 3072, 4096])
def test_other_key_sizes_roundtrip_and_bits(bits):
    nk = NativeKey(seeded_key(bits))
    key, N = nk.key, 130 if bits < 4096 else 66
    m = plaintexts(key, N, bits)
    r_l = orc.synth_r_limbs(4000 + bits, N, key.randbits)
    dm, dr = DevArray(ints_to_limbs(m, nk.nw)), DevArray(r_l)
    ct = DevArray(shape=(N, nk.cw))
    _native.check(nk.lib.pai_encrypt(nk.pk, dm.ptr, dr.ptr, N, ct.ptr, None))
    got = limbs_to_ints(ct.get())
    r = limbs_to_ints(r_l)
    check = range(N) if bits <= 3072 else range(0, N, 4)
    for i in check:
        assert got[i] == orc.encrypt(key, m[i], r[i]), i
    out = DevArray(shape=(N, nk.nw))
    _native.check(nk.lib.pai_decrypt(nk.sk, ct.ptr, N, out.ptr, None))
    assert limbs_to_ints(out.get()) == m


@pytest.mark.parametrize("bits,N", [(1024, 255), (1024, 513), (2048, 1), (2048, 257), (2048, 777), (3072, 256), (3072, 259), (4096, 130)])
def test_tile_boundaries_every_engine(bits, N):
    """Batch lengths around the 256-element tile on every key size (each size runs a different set of engines):
    DJN encrypt bits, decrypt, ct*pt with per-element 53-bit multipliers, inverse — sampled against CPython pow."""
    nk = NativeKey(bench_key() if bits == 2048 else seeded_key(bits))
    key = nk.key
    rng = np.random.default_rng(bits + N)
    m = plaintexts(key, N, bits + N) if N >= 3 else rand_below(rng, key.n, N)
    r_l = orc.synth_r_limbs(7000 + N, N, key.randbits)
    dm, dr = DevArray(ints_to_limbs(m, nk.nw)), DevArray(r_l)
    ct = DevArray(shape=(N, nk.cw))
    _native.check(nk.lib.pai_encrypt(nk.pk, dm.ptr, dr.ptr, N, ct.ptr, None))
    cts = limbs_to_ints(ct.get())
    rs = limbs_to_ints(r_l)
    sample = sorted(set([0, N - 1, N // 2] + [int(i) for i in rng.integers(0, N, 6)]))
    for i in sample:
        assert cts[i] == orc.encrypt(key, m[i], rs[i]), i
    out = DevArray(shape=(N, nk.nw))
    _native.check(nk.lib.pai_decrypt(nk.sk, ct.ptr, N, out.ptr, None))
    assert limbs_to_ints(out.get()) == m
    es = [int(v) | 1 for v in rng.integers(0, 1 << 53, size=N)]
    de = DevArray(ints_to_limbs(es, 2))
    res = DevArray(shape=(N, nk.cw))
    _native.check(nk.lib.pai_ct_mul(nk.pk, ct.ptr, de.ptr, 2, 53, 0, N, res.ptr, None))
    got = limbs_to_ints(res.get())
    for i in sample:
        assert got[i] == pow(cts[i], es[i], key.nsq), i
    _native.check(nk.lib.pai_ct_invert(nk.pk, ct.ptr, N, res.ptr, None))
    got = limbs_to_ints(res.get())
    for i in sample:
        assert got[i] * cts[i] % key.nsq == 1, i


def test_standard_scheme_2048():
    nk = NativeKey(bench_key(djn=False))
    key, N = nk.key, 70
    m = plaintexts(key, N, 9)
    rng = np.random.default_rng(10)
    r = [x + 1 for x in rand_below(rng, key.n - 1, N)]
    dm, dr = DevArray(ints_to_limbs(m, nk.nw)), DevArray(ints_to_limbs(r, nk.nw))
    ct = DevArray(shape=(N, nk.cw))
    _native.check(nk.lib.pai_encrypt(nk.pk, dm.ptr, dr.ptr, N, ct.ptr, None))
    assert limbs_to_ints(ct.get()) == [orc.encrypt(key, x, rr) for x, rr in zip(m, r)]
    out = DevArray(shape=(N, nk.nw))
    _native.check(nk.lib.pai_decrypt(nk.sk, ct.ptr, N, out.ptr, None))
    assert limbs_to_ints(out.get()) == m


@pytest.mark.parametrize("chunk,N", [(None, 1), (None, 77), (8, 1000), (32, 4099), (5, 333)])
def test_ct_invert_2048(k2048, chunk, N, monkeypatch):
    """Batched inversion (product tree + one wave-parallel extended GCD per top-level product; the env hook
    moves the level at which the tree stops) against pow(x, -1, n^2)."""
    import os

    tune(monkeypatch, "invert_chunk", chunk)
    key = k2048.key
    rng = np.random.default_rng(1000 + N)
    a = rand_below(rng, key.nsq, N)
    a[0] = 1
    if N > 3:
        a[1], a[2], a[3] = key.nsq - 1, 2, key.n + 1
    da = DevArray(ints_to_limbs(a, k2048.cw))
    out = DevArray(shape=(N, k2048.cw))
    _native.check(k2048.lib.pai_ct_invert(k2048.pk, da.ptr, N, out.ptr, None))
    got = limbs_to_ints(out.get())
    step = max(1, N // 400)
    for i in list(range(0, N, step)) + [N - 1]:
        assert got[i] == pow(a[i], -1, key.nsq), i
    assert all(0 < g < key.nsq for g in got)


@pytest.mark.parametrize("N,top", [(3, None), (130, 7), (1025, None)])
def test_ct_invert_in_place(k2048, N, top, monkeypatch):
    """d_out == d_ct: the product tree reads both halves of a level after writing one of them, so the library
    stages the leaves."""
    tune(monkeypatch, "invert_chunk", top)
    key = k2048.key
    a = rand_below(np.random.default_rng(77 + N), key.nsq, N)
    da = DevArray(ints_to_limbs(a, k2048.cw))
    _native.check(k2048.lib.pai_ct_invert(k2048.pk, da.ptr, N, da.ptr, None))
    got = limbs_to_ints(da.get())
    step = max(1, N // 100)
    for i in list(range(0, N, step)) + [N - 1]:
        assert got[i] == pow(a[i], -1, key.nsq), i


def test_ct_invert_rejects_non_units(k2048):
    key = k2048.key
    a = [5, key.p * 12345, 7]
    da = DevArray(ints_to_limbs(a, k2048.cw))
    out = DevArray(shape=(3, k2048.cw))
    rc = k2048.lib.pai_ct_invert(k2048.pk, da.ptr, 3, out.ptr, None)
    assert rc == _native.PAI_E_INVALID


def test_ct_invert_other_key_sizes():
    for bits in (1024, 4096):
        nk = NativeKey(seeded_key(bits))
        key, N = nk.key, 40
        rng = np.random.default_rng(bits)
        a = rand_below(rng, key.nsq, N)
        da = DevArray(ints_to_limbs(a, nk.cw))
        out = DevArray(shape=(N, nk.cw))
        _native.check(nk.lib.pai_ct_invert(nk.pk, da.ptr, N, out.ptr, None))
        assert limbs_to_ints(out.get()) == [pow(x, -1, key.nsq) for x in a]


# ---- round 2: tile I/O kernels, single-product trees, reductions in the library -------------------------------
@pytest.mark.parametrize("bits", [1024, 2048, 3072, 4096])
@pytest.mark.parametrize("lat_add", ["0", "4096"])
def test_ct_add_tile_edges_and_broadcast(bits, lat_add, monkeypatch):
    """k_modmul with coalesced tile I/O: ragged last tile, one-element batches, broadcast addend (one Montgomery
    product per element), in-place output — against a*b mod n^2."""
    monkeypatch.setenv("PAI_LAT_ADD_MAX", lat_add)      # small batches: n^2 on the latency geometry (an integer per wavefront) or not
    nk = NativeKey(bench_key() if bits == 2048 else seeded_key(bits))
    key = nk.key
    rng = np.random.default_rng(bits + 1)
    for N in (1, 63, 64, 65, 200, 1031):
        a, b = rand_below(rng, key.nsq, N), rand_below(rng, key.nsq, N)
        a[0], b[0] = key.nsq - 1, key.nsq - 1
        if N > 2:
            a[1], b[1] = 0, 5
            a[2], b[2] = 1, key.nsq - 2
        da, db = DevArray(ints_to_limbs(a, nk.cw)), DevArray(ints_to_limbs(b, nk.cw))
        out = DevArray(shape=(N, nk.cw))
        _native.check(nk.lib.pai_ct_add(nk.pk, da.ptr, db.ptr, 0, N, out.ptr, None))
        assert limbs_to_ints(out.get()) == [x * y % key.nsq for x, y in zip(a, b)]
        _native.check(nk.lib.pai_ct_add(nk.pk, da.ptr, db.ptr, 1, N, out.ptr, None))          # b[0] for everyone
        assert limbs_to_ints(out.get()) == [x * b[0] % key.nsq for x in a]
        _native.check(nk.lib.pai_ct_add(nk.pk, da.ptr, db.ptr, 0, N, da.ptr, None))           # d_out aliases d_a
        assert limbs_to_ints(da.get()) == [x * y % key.nsq for x, y in zip(a, b)]


@pytest.mark.parametrize("bits,members,groups", [(2048, 1, 5), (2048, 2, 1), (2048, 3, 70), (2048, 7, 9), (2048, 64, 3),
                                                  (2048, 1000, 1), (2048, 129, 65), (1024, 37, 4), (3072, 11, 6), (4096, 5, 33)])
def test_ct_prod_matches_plain_products(bits, members, groups):
    """pai_ct_prod: out[g] = prod_l ct[l*groups + g] mod n^2 (member-major) on single Montgomery products with the
    per-level R-power bookkeeping; odd member counts exercise the partner-less nodes."""
    nk = NativeKey(bench_key() if bits == 2048 else seeded_key(bits))
    key = nk.key
    rng = np.random.default_rng(members * 1000 + groups)
    count = members * groups
    a = rand_below(rng, key.nsq, count)
    a[0] = key.nsq - 1
    da = DevArray(ints_to_limbs(a, nk.cw))
    out = DevArray(shape=(groups, nk.cw))
    _native.check(nk.lib.pai_ct_prod(nk.pk, da.ptr, count, groups, out.ptr, None))
    want = []
    for g in range(groups):
        p = 1
        for l in range(members):
            p = p * a[l * groups + g] % key.nsq
        want.append(p)
    assert limbs_to_ints(out.get()) == want
    assert nk.lib.pai_ct_prod(nk.pk, da.ptr, count, groups + count + 1, out.ptr, None) == _native.PAI_E_INVALID


@pytest.mark.parametrize("lat_add", ["0", "4096"])
def test_pow2_tile_skips_and_mixed_deltas(k2048, lat_add, monkeypatch):
    """k_pow2 on tile I/O: tiles whose deltas are all <= 0 are skipped by the workgroup, untouched rows inside a live
    tile are written back unchanged; on the throughput and (small batches, PAI_LAT_ADD_MAX) on the latency geometry."""
    monkeypatch.setenv("PAI_LAT_ADD_MAX", lat_add)
    key, N = k2048.key, 64 * 5 + 9
    rng = np.random.default_rng(4242)
    a = rand_below(rng, key.nsq, N)
    delta = np.zeros(N, dtype=np.int32)
    delta[70] = 3                      # one live element in tile 1
    delta[128:192] = rng.integers(-2, 12, 64)
    delta[N - 1] = 1
    da = DevArray(ints_to_limbs(a, k2048.cw))
    dd = DevArray(delta)
    _native.check(k2048.lib.pai_ct_pow2(k2048.pk, da.ptr, dd.ptr, 0, N, None))
    want = [pow(x, 1 << int(d), key.nsq) if d > 0 else x for x, d in zip(a, delta)]
    assert limbs_to_ints(da.get()) == want
    one = DevArray(np.array([2], dtype=np.int32))
    _native.check(k2048.lib.pai_ct_pow2(k2048.pk, da.ptr, one.ptr, 1, N, None))
    assert limbs_to_ints(da.get()) == [pow(x, 4, key.nsq) for x in want]


def test_fixed_base_table_is_built_lazily_and_calls_keep_the_device():
    """creating a DJN handle must not allocate the multi-GB fixed-base table; the first obfuscating call
    does.  No call changes the thread's current device."""
    hip = C.CDLL("libamdhip64.so")
    free0, total = C.c_size_t(), C.c_size_t()
    nk = None
    hip.hipMemGetInfo(C.byref(free0), C.byref(total))
    nk = NativeKey(bench_key())
    free1 = C.c_size_t()
    hip.hipMemGetInfo(C.byref(free1), C.byref(total))
    assert free0.value - free1.value < (512 << 20), "key creation allocated the fixed-base table"
    key, N = nk.key, 5
    m = plaintexts(key, N, 3)
    ct = DevArray(shape=(N, nk.cw))
    dm = DevArray(ints_to_limbs(m, nk.nw))
    _native.check(nk.lib.pai_raw_encrypt(nk.pk, dm.ptr, N, ct.ptr, None))
    out = DevArray(shape=(N, nk.nw))
    _native.check(nk.lib.pai_decrypt(nk.sk, ct.ptr, N, out.ptr, None))
    assert limbs_to_ints(out.get()) == m
    free2 = C.c_size_t()
    hip.hipMemGetInfo(C.byref(free2), C.byref(total))
    assert free0.value - free2.value < (1024 << 20), "raw encrypt / decrypt built the fixed-base table"
    r = orc.synth_r_limbs(5, N, key.randbits)
    dr = DevArray(r)
    _native.check(nk.lib.pai_encrypt(nk.pk, dm.ptr, dr.ptr, N, ct.ptr, None))
    assert limbs_to_ints(ct.get()) == [orc.encrypt(key, x, rr) for x, rr in zip(m, orc.limbs_to_ints(r))]
    dev = C.c_int(-1)
    hip.hipGetDevice(C.byref(dev))
    assert dev.value == 0


def test_two_streams_share_one_key_handle(k2048):
    """Scratch users of one handle on different streams are chained with events: interleaved decrypt / ct*pt calls on
    two streams give the same bits as serial calls."""
    hip = C.CDLL("libamdhip64.so")
    s1, s2 = C.c_void_p(), C.c_void_p()
    assert hip.hipStreamCreate(C.byref(s1)) == 0 and hip.hipStreamCreate(C.byref(s2)) == 0
    key, N = k2048.key, 600
    rng = np.random.default_rng(31337)
    m1, m2 = plaintexts(key, N, 1), plaintexts(key, N, 2)
    c1 = [orc.raw_encrypt(x, key.n) * pow(key.hs, 3 + i, key.nsq) % key.nsq for i, x in enumerate(m1)]
    c2 = [orc.raw_encrypt(x, key.n) * pow(key.hs, 5 + i, key.nsq) % key.nsq for i, x in enumerate(m2)]
    d1, d2 = DevArray(ints_to_limbs(c1, k2048.cw)), DevArray(ints_to_limbs(c2, k2048.cw))
    o1, o2 = DevArray(shape=(N, k2048.nw)), DevArray(shape=(N, k2048.nw))
    e = [int(v) for v in rng.integers(1, 1 << 40, N)]
    de = DevArray(ints_to_limbs(e, 2))
    p1, p2 = DevArray(shape=(N, k2048.cw)), DevArray(shape=(N, k2048.cw))
    for _ in range(3):
        _native.check(k2048.lib.pai_decrypt(k2048.sk, d1.ptr, N, o1.ptr, s1))
        _native.check(k2048.lib.pai_decrypt(k2048.sk, d2.ptr, N, o2.ptr, s2))
        _native.check(k2048.lib.pai_ct_mul(k2048.pk, d1.ptr, de.ptr, 2, 40, 0, N, p1.ptr, s1))
        _native.check(k2048.lib.pai_ct_mul(k2048.pk, d2.ptr, de.ptr, 2, 40, 0, N, p2.ptr, s2))
    _native.check(k2048.lib.pai_stream_sync(0, s1))
    _native.check(k2048.lib.pai_stream_sync(0, s2))
    assert limbs_to_ints(o1.get()) == m1 and limbs_to_ints(o2.get()) == m2
    assert limbs_to_ints(p1.get())[:50] == [pow(c, x, key.nsq) for c, x in zip(c1[:50], e[:50])]
    assert limbs_to_ints(p2.get())[-50:] == [pow(c, x, key.nsq) for c, x in zip(c2[-50:], e[-50:])]
    hip.hipStreamDestroy(s1)
    hip.hipStreamDestroy(s2)


def test_shard_plan_gather_scatter_roundtrip(k2048):
    """pai_shard_plan / pai_scatter / pai_gather through raw ctypes, with the shards living on device 0 (the only
    device a test box has): the copies and the offsets are the same code that runs across xGMI peers."""
    lib = k2048.lib
    N, W, G = 1003, 8, 3
    host = np.random.default_rng(8).integers(0, 1 << 32, size=(N, W), dtype=np.uint32)
    src = DevArray(host)
    b, c = C.c_size_t(), C.c_size_t()
    plan = []
    for g in range(G):
        _native.check(lib.pai_shard_plan(N, G, g, C.byref(b), C.byref(c)))
        plan.append((b.value, c.value))
    assert plan == [(0, 335), (335, 335), (670, 333)]
    shards = [DevArray(shape=(cnt, W)) for _, cnt in plan]
    devs = (C.c_int32 * G)(0, 0, 0)
    ptrs = (C.c_void_p * G)(*[s.ptr.value for s in shards])
    rows = (C.c_size_t * G)(*[cnt for _, cnt in plan])
    _native.check(lib.pai_scatter(G, devs, ptrs, rows, W, 0, src.ptr))
    for (beg, cnt), s in zip(plan, shards):
        assert np.array_equal(s.get(), host[beg:beg + cnt])
    out = DevArray(shape=(N, W))
    _native.check(lib.pai_gather(G, devs, ptrs, rows, W, 0, out.ptr))
    assert np.array_equal(out.get(), host)
    assert lib.pai_shard_plan(10, 4, 3, C.byref(b), C.byref(c)) == 0 and (b.value, c.value) == (9, 1)
    assert lib.pai_shard_plan(2, 4, 3, C.byref(b), C.byref(c)) == 0 and (b.value, c.value) == (2, 0)


@pytest.mark.parametrize("bits", [1024, 2048, 3072, 4096])
def test_decrypt_latency_and_throughput_paths_agree(bits, monkeypatch):
    """Small batches decrypt on the wide-group geometries (an integer spread over 16-64 lanes: latency path), large
    ones on the digit-pair engine (throughput path); PAI_LATENCY_MAX moves the switch.  Both against the oracle."""
    nk = NativeKey(bench_key() if bits == 2048 else seeded_key(bits))
    key = nk.key
    for N in ((3, 7, 33, 150, 700) if bits <= 2048 else (3, 33, 70, 600)):      # > 512: the denser wide-group geometry
        m = plaintexts(key, N, bits + N)
        rng = np.random.default_rng(N)
        # short randomness keeps the oracle's CPython pow cheap; decryption does not care how a ciphertext was obfuscated
        ct = [orc.encrypt(key, x, int.from_bytes(rng.bytes(16), "little")) for x in m]
        dct = DevArray(ints_to_limbs(ct, nk.cw))
        # PAI_TUNE lat_rl: up to one integer per CU the latency path runs right to left on wave pairs (k_dec_a_rl: squarings on
        # one wave, products on another); 0 keeps the left-to-right window kernel
        # PAI_TUNE lat_pp: the smallest batches (one workgroup per (ciphertext, prime)) run stage A on digit pairs with base
        # s k, pipelined over four waves (k_dec_a_pp); 0 leaves them to the wave-pair / window kernels
        for switch, rl, pp in (("0", "100000", "0"), ("100000", "100000", "100000"), ("100000", "100000", "0"), ("100000", "0", "0")):
            monkeypatch.setenv("PAI_LATENCY_MAX", switch)
            tune(monkeypatch, "lat_rl", rl)
            tune(monkeypatch, "lat_pp", pp)
            out = DevArray(shape=(N, nk.nw))
            _native.check(nk.lib.pai_decrypt(nk.sk, dct.ptr, N, out.ptr, None))
            assert limbs_to_ints(out.get()) == m, (bits, N, switch, rl, pp)
        if True:
            # PAI_TUNE dec_mid_min / dec_mid_max: mid-size batches (2 049 ... 18 432 at 2048-bit keys) run stage A as the
            # lane-group digit-pair exponentiation with modulus s (k_pair_ctmul, 4 lanes x 9 / 14 / 18 limbs, both primes in one launch)
            tune(monkeypatch, "dec_mid_min", 0)
            tune(monkeypatch, "dec_mid_max", 1 << 30)
            out = DevArray(shape=(N, nk.nw))
            _native.check(nk.lib.pai_decrypt(nk.sk, dct.ptr, N, out.ptr, None))
            assert limbs_to_ints(out.get()) == m, (bits, N, "mid")
            tune(monkeypatch, "dec_mid_min", None)
            tune(monkeypatch, "dec_mid_max", None)


@pytest.mark.parametrize("bits", [1024, 2048, 4096])
def test_ct_mul_latency_and_throughput_paths_agree(bits, monkeypatch):
    nk = NativeKey(bench_key() if bits == 2048 else seeded_key(bits))
    key = nk.key
    rng = np.random.default_rng(bits)
    for N, ebits in ((5, 53), (40, 12), (64, 300), (3, 9), (300, 64)):
        c = rand_below(rng, key.nsq, N)
        e = [int.from_bytes(rng.bytes(ebits // 8 + 1), "little") % (1 << ebits) for _ in range(N)]
        e[0] = 0
        e[2] = 1
        c[1] = key.nsq - 1
        if N > 4:
            e[3], e[4] = (1 << ebits) - 1, 1 << (ebits - 1)
        ew = (ebits + 31) // 32
        dc, de = DevArray(ints_to_limbs(c, nk.cw)), DevArray(ints_to_limbs(e, ew))
        # PAI_TUNE lat_mul_pp: the smallest batches (keys up to 2048 bits) run on digit pairs with base n k, one workgroup per
        # ciphertext and the chain pipelined over its four waves (k_ctmul_pp); 0 leaves them to the kernels below
        # PAI_TUNE lat_mul_rl: right to left on wave pairs (k_modexp_rl, no table); 0 = windowed kernel
        want = pow_many(c, e, key.nsq)
        for switch, pp, rl in (("0", "0", "100000"), ("100000", "100000", "100000"), ("100000", "0", "100000"), ("100000", "0", "0")):
            monkeypatch.setenv("PAI_LATENCY_MAX", switch)
            tune(monkeypatch, "lat_mul_pp", pp)
            tune(monkeypatch, "lat_mul_rl", rl)
            out = DevArray(shape=(N, nk.cw))
            _native.check(nk.lib.pai_ct_mul(nk.pk, dc.ptr, de.ptr, ew, ebits, 0, N, out.ptr, None))
            assert limbs_to_ints(out.get()) == want, (bits, N, ebits, switch, pp, rl)
        if bits <= 2048:
            # PAI_TUNE ctmul_mid_min / ctmul_mid_max: mid-size batches (5 120 ... 49 152) at keys the one-element-per-lane engine serves
            # run the lane-group digit-pair exponentiation on 4 lanes per ciphertext (k_pair_ctmul with n on 36 / 72 limbs)
            tune(monkeypatch, "ctmul_mid_min", 0)
            tune(monkeypatch, "ctmul_mid_max", 1 << 30)
            out = DevArray(shape=(N, nk.cw))
            _native.check(nk.lib.pai_ct_mul(nk.pk, dc.ptr, de.ptr, ew, ebits, 0, N, out.ptr, None))
            assert limbs_to_ints(out.get()) == want, (bits, N, ebits, "mid")
            tune(monkeypatch, "ctmul_mid_min", None)
            tune(monkeypatch, "ctmul_mid_max", None)
        # one broadcast exponent
        for pp in ("100000", "0"):
            tune(monkeypatch, "lat_mul_pp", pp)
            tune(monkeypatch, "lat_mul_rl", 100000)
            out = DevArray(shape=(N, nk.cw))
            _native.check(nk.lib.pai_ct_mul(nk.pk, dc.ptr, C.c_void_p(de.ptr.value + 4 * ew), ew, ebits, 1, N, out.ptr, None))
            assert limbs_to_ints(out.get()) == pow_many(c, e[1], key.nsq), (bits, N, ebits, "bcast", pp)


@pytest.mark.parametrize("bits", [1024, 2048, 3072, 4096])
def test_djn_encrypt_latency_and_throughput_paths_agree(bits, monkeypatch):
    """Small DJN batches encrypt (and re-obfuscate) on the wide-group geometry with its own 10-bit fixed-base table."""
    nk = NativeKey(bench_key() if bits == 2048 else seeded_key(bits))
    key = nk.key
    for N in ((3, 20, 65) if bits <= 2048 else (3, 37)):          # the oracle's CPython pow dominates at the wide keys
        m = plaintexts(key, N, bits + 3 * N)
        r = orc.synth_r_limbs(bits + N, N, key.randbits)
        r[0] = 0                                                   # r = 0: obfuscator 1
        r[1] = 0xFFFFFFFF
        r[1, -1] &= np.uint32((1 << (key.randbits - 32 * (r.shape[1] - 1))) - 1) if key.randbits % 32 else np.uint32(0xFFFFFFFF)
        r_int = orc.limbs_to_ints(r)
        want = djn_encrypt_many(key, m, r_int)                     # the oracle's formula; bulk powers through the C oracle
        assert want[:2] == [orc.encrypt(key, x, rr) for x, rr in zip(m[:2], r_int[:2])]
        want2 = djn_obfuscate_many(key, want, r_int)
        dm, dr = DevArray(ints_to_limbs(m, nk.nw)), DevArray(r)
        # PAI_TUNE lat_enc_tree: the four waves of a workgroup share one wave's integers (k_encrypt_tree); 0 = one chain per integer
        # PAI_DISABLE lat_enc_m1: the shared chain on a minus-one context of n^2 (table converted once) or on the conventional one
        for switch, tree, m1 in (("0", "100000", "1"), ("100000", "100000", "1"), ("100000", "100000", "0"), ("100000", "0", "1")):
            monkeypatch.setenv("PAI_LATENCY_MAX", switch)
            tune(monkeypatch, "lat_enc_tree", tree)
            knob_disable(monkeypatch, "lat_enc_m1", m1 == "0")
            ct = DevArray(shape=(N, nk.cw))
            _native.check(nk.lib.pai_encrypt(nk.pk, dm.ptr, dr.ptr, N, ct.ptr, None))
            assert limbs_to_ints(ct.get()) == want, (bits, N, switch, tree, m1)
            _native.check(nk.lib.pai_obfuscate(nk.pk, ct.ptr, dr.ptr, N, None))
            assert limbs_to_ints(ct.get()) == want2, (bits, N, switch, tree, m1)
        if bits <= 2048:
            # PAI_TUNE enc_mid_min / enc_mid_max: mid-size batches (4 096 ... 40 960) at keys the one-element-per-lane engine serves run
            # the lane-group digit-pair kernel with 4 lanes per element on that engine's own fixed-base table (same layout, same R)
            tune(monkeypatch, "enc_mid_min", 0)
            tune(monkeypatch, "enc_mid_max", 1 << 30)
            ct = DevArray(shape=(N, nk.cw))
            _native.check(nk.lib.pai_encrypt(nk.pk, dm.ptr, dr.ptr, N, ct.ptr, None))
            assert limbs_to_ints(ct.get()) == want, (bits, N, "mid")
            _native.check(nk.lib.pai_obfuscate(nk.pk, ct.ptr, dr.ptr, N, None))
            assert limbs_to_ints(ct.get()) == want2, (bits, N, "mid")
            tune(monkeypatch, "enc_mid_min", None)
            tune(monkeypatch, "enc_mid_max", None)


@pytest.mark.parametrize("bits", [1024, 2048, 3072, 4096])
@pytest.mark.parametrize("lat_add", ["0", "4096"])
def test_ct_add_aligned_matches_the_two_step_definition(bits, lat_add, monkeypatch):
    """pai_ct_add_aligned: the lower-exponent side is raised by ^(2^|delta|), then the ciphertexts are multiplied
    (ipcl_python.py:570-741 + :490-526) — one kernel, against CPython pow, with zero / positive / negative deltas
    mixed inside wave tiles, a broadcast right operand and in-place output; on the throughput geometry and (small
    batches, PAI_LAT_ADD_MAX) on the latency geometry."""
    monkeypatch.setenv("PAI_LAT_ADD_MAX", lat_add)
    nk = NativeKey(bench_key() if bits == 2048 else seeded_key(bits))
    key, M = nk.key, nk.key.nsq
    rng = np.random.default_rng(bits + 7)
    for N in (1, 17, 130):
        a, b = rand_below(rng, M, N), rand_below(rng, M, N)
        delta = rng.integers(-6, 7, N).astype(np.int32)
        delta[0] = 0
        if N > 3:
            delta[1], delta[2], delta[3] = 9, -11, 0
        da, db, dd = DevArray(ints_to_limbs(a, nk.cw)), DevArray(ints_to_limbs(b, nk.cw)), DevArray(delta)
        out = DevArray(shape=(N, nk.cw))
        _native.check(nk.lib.pai_ct_add_aligned(nk.pk, da.ptr, db.ptr, 0, dd.ptr, N, out.ptr, None))
        want = [x * pow(y, 1 << int(d), M) % M if d > 0 else pow(x, 1 << int(-d), M) * y % M for x, y, d in zip(a, b, delta)]
        assert limbs_to_ints(out.get()) == want, (bits, N)
        _native.check(nk.lib.pai_ct_add_aligned(nk.pk, da.ptr, db.ptr, 1, dd.ptr, N, out.ptr, None))
        want_b = [x * pow(b[0], 1 << int(d), M) % M if d > 0 else pow(x, 1 << int(-d), M) * b[0] % M for x, d in zip(a, delta)]
        assert limbs_to_ints(out.get()) == want_b, (bits, N, "bcast")
        _native.check(nk.lib.pai_ct_add_aligned(nk.pk, da.ptr, db.ptr, 0, dd.ptr, N, da.ptr, None))
        assert limbs_to_ints(da.get()) == want, (bits, N, "in place")


def test_buf_slice_and_rotate_are_row_copies():
    """pai_buf_slice / pai_buf_rotate (SURVEY 8b's container exports; classes.cpp:224-262,328-366): slices with a step,
    rotations by any signed amount, against numpy."""
    lib = _native.load()
    rng = np.random.default_rng(77)
    N, W = 37, 13
    src = rng.integers(0, 1 << 32, size=(N, W), dtype=np.uint64).astype(np.uint32)
    ds = DevArray(src)
    for start, count, step in ((0, N, 1), (5, 20, 1), (3, 11, 3), (36, 1, 1), (0, 0, 1), (1, 18, 2)):
        out = DevArray(shape=(max(count, 1), W))
        _native.check(lib.pai_buf_slice(0, ds.ptr, W, start, count, step, out.ptr, None))
        _native.check(lib.pai_stream_sync(0, None))
        assert np.array_equal(out.get()[:count], src[start:start + count * step:step][:count])
    for shift in (0, 1, 5, N - 1, N, N + 3, -1, -40):
        out = DevArray(shape=(N, W))
        _native.check(lib.pai_buf_rotate(0, ds.ptr, W, N, shift, out.ptr, None))
        _native.check(lib.pai_stream_sync(0, None))
        assert np.array_equal(out.get(), np.roll(src, -shift, axis=0)), shift
    assert lib.pai_buf_rotate(0, ds.ptr, W, N, 1, ds.ptr, None) != 0          # in place is refused


@pytest.mark.parametrize("bits,R,K,M,lanes,wbits", [(2048, 2, 37, 5, None, None), (2048, 1, 64, 3, "40", "5"), (1024, 3, 9, 4, "7", "3"),
                                                    (2048, 1, 1, 1, None, "1"), (2048, 2, 11, 7, "9", "7"), (1024, 1, 20, 2, "5", "6"),
                                                    (3072, 2, 13, 3, "6", None), (4096, 1, 9, 5, "4", "5"), (3072, 1, 3, 2, None, "2")])
def test_ct_multiexp_matches_the_product_of_powers(bits, R, K, M, lanes, wbits, monkeypatch):
    """pai_ct_multiexp: out[r*M + j] = prod_l base(r, l, j)^e[r][l][j] with the inverse's table where the sign byte is
    set; exponents of up to 75 bits with zero windows, zeros and ones; PAI_TUNE mexp_lanes forces chunks of several members
    (shared squarings) on these small shapes, PAI_TUNE mexp_wbits the table width (windows that straddle exponent words)."""
    if lanes is not None:
        tune(monkeypatch, "mexp_lanes", lanes)
    if wbits is not None:
        tune(monkeypatch, "mexp_wbits", wbits)                     # the default follows the shape (2 .. 7 bits)
    nk = NativeKey(bench_key() if bits == 2048 else seeded_key(bits))
    key = nk.key
    rng = np.random.default_rng(1000 * R + K)
    base = rand_below(rng, key.nsq, R * K)
    base = [b if (b % key.p and b % key.q) else 3 for b in base]
    inv = [pow(b, -1, key.nsq) for b in base]
    ew = 3
    e = [[[int.from_bytes(rng.bytes(10), "little") >> int(rng.integers(5, 80)) for _ in range(M)] for _ in range(K)] for _ in range(R)]
    e[0][0][0] = 0
    if K > 1:
        e[0][1][0] = 1
    sign = rng.integers(0, 2, size=(K, M)).astype(np.uint8)
    ebits = max(1, max(v.bit_length() for a in e for b in a for v in b))
    e_l = np.zeros((R, K, M, ew), dtype=np.uint32)
    for r in range(R):
        for l in range(K):
            for j in range(M):
                for w in range(ew):
                    e_l[r, l, j, w] = (e[r][l][j] >> (32 * w)) & 0xFFFFFFFF
    want = []
    for r in range(R):
        for j in range(M):
            acc = 1
            for l in range(K):
                b = inv[r * K + l] if sign[l, j] else base[r * K + l]
                acc = acc * pow(b, e[r][l][j], key.nsq) % key.nsq
            want.append(acc)
    dc, di, de, dsg = DevArray(ints_to_limbs(base, nk.cw)), DevArray(ints_to_limbs(inv, nk.cw)), DevArray(e_l), DevArray(sign)
    out = DevArray(shape=(R * M, nk.cw))
    _native.check(nk.lib.pai_ct_multiexp(nk.pk, dc.ptr, di.ptr, R, K, M, de.ptr, ew, ebits, dsg.ptr, out.ptr, None))
    assert limbs_to_ints(out.get()) == want
    # without signs: every base is the ciphertext itself
    _native.check(nk.lib.pai_ct_multiexp(nk.pk, dc.ptr, None, R, K, M, de.ptr, ew, ebits, None, out.ptr, None))
    want0 = []
    for r in range(R):
        for j in range(M):
            acc = 1
            for l in range(K):
                acc = acc * pow(base[r * K + l], e[r][l][j], key.nsq) % key.nsq
            want0.append(acc)
    assert limbs_to_ints(out.get()) == want0


@pytest.mark.parametrize("bits", [1024, 2048, 3072, 4096])
@pytest.mark.parametrize("lat_add", ["0", "4096"])
def test_lazy_montgomery_domain_exports(bits, lat_add, monkeypatch):
    """pai_ct_mont_mul / pai_pubkey_mont_bits / pai_ct_add_aligned_dom (include/paillier_hip.h, 'lazy Montgomery domain'):
    the single product a b R^-1 against CPython ints, the tag algebra (ka, kb -> ka + kb - 1; retag by a broadcast
    constant), a chain of three single-product additions brought back to the wire form with one more product = the bits of
    three pai_ct_add calls, and the aligned addition on operands stored as x R^k for k in {-2, -1, 1, 3}.  Small batches
    run the single product on the latency geometry with a constant that keeps R the throughput geometry's (PAI_LAT_ADD_MAX)."""
    monkeypatch.setenv("PAI_LAT_ADD_MAX", lat_add)
    nk = NativeKey(bench_key() if bits == 2048 else seeded_key(bits))
    M = nk.key.nsq
    rb = C.c_int(0)
    _native.check(nk.lib.pai_pubkey_mont_bits(nk.pk, C.byref(rb)))
    assert rb.value % 29 == 0 and (1 << rb.value) > 4 * M
    R = pow(2, rb.value, M)
    Ri = pow(R, -1, M)
    rng = np.random.default_rng(bits + 31)

    def rk(k):
        return pow(R, k, M) if k >= 0 else pow(Ri, -k, M)

    for N in (1, 17, 130):
        a, b, c, d = (rand_below(rng, M, N) for _ in range(4))
        a[0] = M - 1
        da, db, dc, dd = (DevArray(ints_to_limbs(v, nk.cw)) for v in (a, b, c, d))
        out = DevArray(shape=(N, nk.cw))
        _native.check(nk.lib.pai_ct_mont_mul(nk.pk, da.ptr, db.ptr, 0, N, out.ptr, None))
        assert limbs_to_ints(out.get()) == [x * y * Ri % M for x, y in zip(a, b)], (bits, N)
        _native.check(nk.lib.pai_ct_mont_mul(nk.pk, da.ptr, db.ptr, 1, N, out.ptr, None))
        assert limbs_to_ints(out.get()) == [x * b[0] * Ri % M for x in a], (bits, N, "bcast")
        # ((a + b) + c) + d with one product each: tag -3; one product with R^4 returns to the wire form
        _native.check(nk.lib.pai_ct_mont_mul(nk.pk, da.ptr, db.ptr, 0, N, out.ptr, None))
        _native.check(nk.lib.pai_ct_mont_mul(nk.pk, out.ptr, dc.ptr, 0, N, out.ptr, None))       # in place
        _native.check(nk.lib.pai_ct_mont_mul(nk.pk, out.ptr, dd.ptr, 0, N, out.ptr, None))
        k4 = DevArray(ints_to_limbs([rk(4)], nk.cw))
        _native.check(nk.lib.pai_ct_mont_mul(nk.pk, out.ptr, k4.ptr, 1, N, out.ptr, None))
        assert limbs_to_ints(out.get()) == [w * x * y * z % M for w, x, y, z in zip(a, b, c, d)], (bits, N, "chain")
        # aligned addition on a common tag k
        delta = rng.integers(-5, 6, N).astype(np.int32)
        delta[0] = 0
        ddl = DevArray(delta)
        plain = [x * pow(y, 1 << int(t), M) % M if t > 0 else pow(x, 1 << int(-t), M) * y % M for x, y, t in zip(a, b, delta)]
        for k in (-2, -1, 1, 3):
            ak, bk = [x * rk(k) % M for x in a], [y * rk(k) % M for y in b]
            dak, dbk = DevArray(ints_to_limbs(ak, nk.cw)), DevArray(ints_to_limbs(bk, nk.cw))
            ent = DevArray(ints_to_limbs([rk(2 - k)], nk.cw))
            _native.check(nk.lib.pai_ct_add_aligned_dom(nk.pk, dak.ptr, dbk.ptr, 0, ddl.ptr, N, out.ptr, ent.ptr, None))
            assert limbs_to_ints(out.get()) == [v * rk(k) % M for v in plain], (bits, N, k)


def test_pubkey_trim_releases_the_tables_and_the_next_call_rebuilds_them(k2048):
    """pai_pubkey_trim (many-key deployments): after a DJN encryption the handle holds its fixed-base table (gigabytes at the
    default window width); trim returns that memory, a second key's table fits next to the first, and the next encryption
    rebuilds the table and produces the same bits."""
    nk, key = k2048, k2048.key
    N = 16384                                                     # beyond the latency path (<= 11 776 at 2048-bit keys): the throughput kernel and its table
    m = plaintexts(key, N, 4242)
    r = orc.synth_r_limbs(4243, N, key.randbits)
    want = [orc.encrypt(key, x, rr) for x, rr in zip(m[:8], orc.limbs_to_ints(r[:8]))]
    dm, dr = DevArray(ints_to_limbs(m, nk.nw)), DevArray(r)
    ct = DevArray(shape=(N, nk.cw))
    _native.check(nk.lib.pai_encrypt(nk.pk, dm.ptr, dr.ptr, N, ct.ptr, None))
    first = ct.get()
    assert limbs_to_ints(first[:8]) == want
    freed = C.c_size_t(0)
    _native.check(nk.lib.pai_pubkey_trim(nk.pk, C.byref(freed)))
    assert freed.value >= 1 << 30, freed.value                   # 8.6 GB at the default 18-bit windows
    other = NativeKey(seeded_key(1024))                           # a second DJN key on the same device builds its own table
    m2 = plaintexts(other.key, 64, 1)
    r2 = orc.synth_r_limbs(2, 64, other.key.randbits)
    ct2 = DevArray(shape=(64, other.cw))
    dm2, dr2 = DevArray(ints_to_limbs(m2, other.nw)), DevArray(r2)
    _native.check(other.lib.pai_encrypt(other.pk, dm2.ptr, dr2.ptr, 64, ct2.ptr, None))
    assert limbs_to_ints(ct2.get()[:2]) == [orc.encrypt(other.key, x, rr) for x, rr in zip(m2[:2], orc.limbs_to_ints(r2[:2]))]
    _native.check(nk.lib.pai_encrypt(nk.pk, dm.ptr, dr.ptr, N, ct.ptr, None))       # rebuilds
    assert np.array_equal(ct.get(), first)
    _native.check(nk.lib.pai_pubkey_trim(nk.pk, None))


def test_async_invert_and_sticky_status():
    """pai_ct_invert_async queues the same work without a synchronisation; a non-unit sets bit 0 of the handle's sticky
    status word (pai_pubkey_status), which the API layer reads before anything leaves the device; pai_ct_pow2_hint with
    a hint below a shift of its batch sets bit 1 instead of returning truncated powers silently."""
    import ctypes as C

    import torch

    from pailliercryptolib_python_amd import _native, engine

    key = orc.make_key(orc.BENCH_P, orc.BENCH_Q, djn_x=0x1234567, bits=2048)
    pub = engine.PublicKeyHandle(key.n, 2048, key.hs, key.randbits, device="cuda:0")
    N = 300
    rng = np.random.default_rng(5)
    vals = [int.from_bytes(rng.bytes(500), "little") % key.nsq | 1 for _ in range(N)]
    ct = engine.to_device_words(engine.ints_to_words(vals, pub.ct_words), pub.device)
    got = pub.ct_invert(ct, sync=False)
    pub.check_status()                                              # units only: nothing to report
    assert engine.words_to_ints(engine.to_host_words(got)) == [pow(v, -1, key.nsq) for v in vals]
    bad = list(vals)
    bad[17] = key.p * 12345                                         # shares a factor with n
    ctb = engine.to_device_words(engine.ints_to_words(bad, pub.ct_words), pub.device)
    pub.ct_invert(ctb, sync=False)                                  # returns; the failure is remembered
    with pytest.raises(_native.NativeError, match="not invertible"):
        pub.check_status()
    pub.check_status(force=True)                                    # cleared by the read
    with pytest.raises(_native.NativeError, match="not invertible"):
        pub.ct_invert(ctb)                                          # the synchronous form still fails at the call
    # an under-estimated pow2 hint on the digit-engine path (batch >= PAI_POW2_DIGIT_MIN)
    M = 1 << 14
    big = ct[:1].expand(M, -1).contiguous()
    d = torch.full((M,), 9, dtype=torch.int32, device=pub.device)
    d[5] = 20
    _native.check(pub.lib.pai_ct_pow2_hint(pub.h, big.data_ptr(), d.data_ptr(), 0, M, 12, None))
    st = C.c_int(0)
    _native.check(pub.lib.pai_pubkey_status(pub.h, C.byref(st), 1, None))
    assert st.value & 2
    big = ct[:1].expand(M, -1).contiguous()
    _native.check(pub.lib.pai_ct_pow2_hint(pub.h, big.data_ptr(), d.data_ptr(), 0, M, 20, None))
    _native.check(pub.lib.pai_pubkey_status(pub.h, C.byref(st), 1, None))
    assert st.value == 0
    rows = engine.words_to_ints(engine.to_host_words(big[[0, 5]]))
    assert rows == [pow(vals[0], 1 << 9, key.nsq), pow(vals[0], 1 << 20, key.nsq)]
    # a hint BELOW the digit path's range runs the lane-group kernel, which serves any shift: correct powers, nothing flagged
    # (the word used to be set for correct ciphertexts)
    small = ct[:64].contiguous()
    d2 = torch.full((M,), 3, dtype=torch.int32, device=pub.device)
    d2[5] = 6
    big = small[:1].expand(M, -1).contiguous()
    _native.check(pub.lib.pai_ct_pow2_hint(pub.h, big.data_ptr(), d2.data_ptr(), 0, M, 4, None))
    _native.check(pub.lib.pai_pubkey_status(pub.h, C.byref(st), 1, None))
    assert st.value == 0
    got = engine.words_to_ints(engine.to_host_words(big[[0, 5]]))
    assert got == [pow(vals[0], 1 << 3, key.nsq), pow(vals[0], 1 << 6, key.nsq)]
    # pai_ct_invert_flag: the outcome goes to the caller's word, not to the handle
    f_ok, f_bad = pub.new_flag(), pub.new_flag()
    got = pub.ct_invert(ct, flag=f_ok)
    pub.ct_invert(ctb, flag=f_bad)
    assert int(f_ok.item()) == 0 and int(f_bad.item()) & 1
    assert engine.words_to_ints(engine.to_host_words(got)) == [pow(v, -1, key.nsq) for v in vals]
    _native.check(pub.lib.pai_pubkey_status(pub.h, C.byref(st), 1, None))
    assert st.value == 0


def test_fixed_base_table_cache_evicts_least_recently_used(monkeypatch):
    """Per-device LRU of the DJN fixed-base tables under a byte budget (PAI_FB_CACHE_MB): with room for one table, a
    second key's first obfuscating call returns the first key's table; the first key rebuilds on its next call and
    produces the same ciphertext bits; device memory in use stays bounded; trim()/destroy keep the list consistent."""
    import torch

    from pailliercryptolib_python_amd import engine

    monkeypatch.setenv("PAI_FB_TABLE_MB", "256")
    monkeypatch.setenv("PAI_FB_CACHE_MB", "400")
    fx = json.loads((Path(__file__).parent / "golden" / "fixture_keys.json").read_text())
    keys = [orc.make_key(int(fx["1024"]["p"], 16), int(fx["1024"]["q"], 16), djn_x=x, bits=1024) for x in (3, 5, 7)]
    pubs = [engine.PublicKeyHandle(k.n, 1024, k.hs, k.randbits, device="cuda:0") for k in keys]
    N = 5000                                                        # beyond the small-batch path: the big tables are used
    m = engine.to_device_words(ints_to_limbs([12345 + i for i in range(N)], pubs[0].n_words), pubs[0].device)
    r_l = orc.synth_r_limbs(5, N, keys[0].randbits)
    r = engine.to_device_words(r_l, pubs[0].device)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    first = [None, None, None]
    for rnd in range(2):
        for i, pub in enumerate(pubs):
            ct = pub.encrypt(m, r)
            torch.cuda.synchronize()
            if first[i] is None:
                first[i] = ct.clone()
                want = orc.encrypt(keys[i], 12345, orc.limbs_to_ints(r_l[:1])[0])
                assert engine.words_to_ints(engine.to_host_words(ct[:1]))[0] == want
            else:
                assert torch.equal(ct, first[i])                    # rebuilt table, same bits
            used = free0 - torch.cuda.mem_get_info()[0]
            assert used < 700 * (1 << 20), f"tables of evicted keys are still resident: {used >> 20} MiB"
    assert pubs[0].trim() >= 0
    ct = pubs[0].encrypt(m, r)
    assert torch.equal(ct, first[0])
    del pubs


@pytest.mark.parametrize("wbits", [5, 10, 14])
def test_small_batch_table_widths_give_the_same_bits(wbits, monkeypatch):
    """PAI_TUNE lat_fb_wbits: the window width of the small-batch DJN table (default 12 bits) only trades memory for latency."""
    tune(monkeypatch, "lat_fb_wbits", wbits)
    monkeypatch.setenv("PAI_LATENCY_MAX", "100000")
    nk = NativeKey(bench_key())
    key = nk.key
    N = 21
    m = plaintexts(key, N, 77 + wbits)
    r = orc.synth_r_limbs(500 + wbits, N, key.randbits)
    want = [orc.encrypt(key, x, rr) for x, rr in zip(m, orc.limbs_to_ints(r))]
    dm, dr, ct = DevArray(ints_to_limbs(m, nk.nw)), DevArray(r), DevArray(shape=(N, nk.cw))
    _native.check(nk.lib.pai_encrypt(nk.pk, dm.ptr, dr.ptr, N, ct.ptr, None))
    assert limbs_to_ints(ct.get()) == want


@pytest.mark.parametrize("bits,env", [(1024, {}), (2048, {"fb_digit_wbits": 4}), (2048, {"fb_digit_wbits": 12}), (2048, {}),
                                      (3072, {"fb_wbits": 6}), (3072, {}), (4096, {"fb_wbits": 5}), (4096, {})])
def test_g_factored_tables_give_the_bits_of_the_plain_tables(bits, env, monkeypatch):
    """Round 4: the fixed-base tables hold g-factored entries (a, t) — x R == a (1 + n)^t — so that a table product is
    the 4 NL^2 rule and the exponents are summed on the side (kernels_padic_enc.hpp / kernels_pair.hpp; conversion by
    simultaneous inversion: k_fb_g_prefix / k_fb_g_finish, k_pair_g_*).  Same ciphertext bits as the plain table
    (PAI_DISABLE=gform) and as the oracle, for encryption and apply_obfuscator, over table widths that make the inversion
    chunks 16 / 32 / 64 entries long, r = 0 and r = all ones included."""
    for k, v in env.items():
        tune(monkeypatch, k, v)
    monkeypatch.setenv("PAI_LATENCY_MAX", "0")                      # the big tables for every batch size
    key = (bench_key() if bits == 2048 else seeded_key(bits))
    N = 70 if bits <= 2048 else 20
    m = plaintexts(key, N, bits + 17)
    r = orc.synth_r_limbs(bits + 5, N, key.randbits)
    r[0] = 0
    r[1] = 0xFFFFFFFF
    r[1, -1] &= np.uint32((1 << (key.randbits - 32 * (r.shape[1] - 1))) - 1) if key.randbits % 32 else np.uint32(0xFFFFFFFF)
    got = {}
    for g in ("1", "0"):
        knob_disable(monkeypatch, "gform", g == "0")
        nk = NativeKey(key)
        dm, dr = DevArray(ints_to_limbs(m, nk.nw)), DevArray(r)
        ct = DevArray(shape=(N, nk.cw))
        _native.check(nk.lib.pai_encrypt(nk.pk, dm.ptr, dr.ptr, N, ct.ptr, None))
        enc = ct.get().copy()
        _native.check(nk.lib.pai_obfuscate(nk.pk, ct.ptr, dr.ptr, N, None))
        got[g] = (enc, ct.get().copy())
        del nk
    assert np.array_equal(got["1"][0], got["0"][0]) and np.array_equal(got["1"][1], got["0"][1])
    rs = orc.limbs_to_ints(r)
    nchk = N if bits <= 2048 else 6
    want = [orc.encrypt(key, x, rr) for x, rr in zip(m[:nchk], rs[:nchk])]
    assert limbs_to_ints(got["1"][0][:nchk]) == want
    assert limbs_to_ints(got["1"][1][:nchk]) == [orc.apply_obfuscator(key, c, rr) for c, rr in zip(want, rs[:nchk])]


def _addn_call(nk, ops, raises, tag0, tag, dom_out, N, out):
    k = len(ops)
    ptrs = (C.c_void_p * k)(*[o.ptr.value for o in ops])
    rz = None
    if raises is not None:
        rz = (C.c_void_p * k)(*[None if r is None else r.ptr.value for r in raises])
    return nk.lib.pai_ct_addn(nk.pk, ptrs, rz, k, tag0, tag, dom_out, N, out.ptr, None)


@pytest.mark.parametrize("bits,k,N", [(2048, 8, 1000), (2048, 2, 17), (2048, 16, 130), (1024, 3, 300), (3072, 5, 100), (4096, 4, 70)])
def test_ct_addn_is_the_product_of_its_operands(bits, k, N):
    """pai_ct_addn (the n-ary ciphertext sum, ipcl_python.py:365-381 as one pass): out = prod_j op_j mod n^2 in the wire form,
    in every lazy-domain arrangement (operand tags, result tag), with per-operand exponent raises (op_j^(2^raise_j), the
    alignment of :570-741), on ragged sizes and with the output aliasing an operand — against CPython integers."""
    key = bench_key() if bits == 2048 else seeded_key(bits)
    nk = NativeKey(key)
    rng = np.random.default_rng(bits + k)
    vals = [rand_below(rng, key.nsq, N) for _ in range(k)]
    ops = [DevArray(ints_to_limbs(v, nk.cw)) for v in vals]
    out = DevArray(shape=(N, nk.cw))
    bits_r = C.c_int(0)
    _native.check(nk.lib.pai_pubkey_mont_bits(nk.pk, C.byref(bits_r)))
    R = pow(2, bits_r.value, key.nsq)

    def rpow(m):
        return pow(R, m, key.nsq) if m >= 0 else pow(pow(R, -1, key.nsq), -m, key.nsq)

    want = [1] * N
    for v in vals:
        want = [a * b % key.nsq for a, b in zip(want, v)]
    # wire form in, wire form out
    _native.check(_addn_call(nk, ops, None, 0, 0, 0, N, out))
    assert limbs_to_ints(out.get()) == want
    # the natural tag: operands x R^0, result (prod) R^(1-k) with no fix-up product
    _native.check(_addn_call(nk, ops, None, 0, 0, 1 - k, N, out))
    assert limbs_to_ints(out.get()) == [w * rpow(1 - k) % key.nsq for w in want]
    # tagged operands: operand 0 holds x R^-2, the others x R^1; result asked at tag 3
    t_ops = [DevArray(ints_to_limbs([x * rpow(-2 if j == 0 else 1) % key.nsq for x in v], nk.cw)) for j, v in enumerate(vals)]
    _native.check(_addn_call(nk, t_ops, None, -2, 1, 3, N, out))
    assert limbs_to_ints(out.get()) == [w * rpow(3) % key.nsq for w in want]
    # exponent raises: some operands without, some with zeros only, some mixed (a whole tile without raise, ragged maxima)
    rz_h = []
    for j in range(k):
        if j % 3 == 0:
            rz_h.append(None)
        elif j % 3 == 1:
            r_ = rng.integers(0, 4, N).astype(np.int32)
            r_[: min(N, 40)] = 0
            rz_h.append(r_)
        else:
            rz_h.append(np.zeros(N, dtype=np.int32) if j > 3 else rng.integers(0, 7, N).astype(np.int32))
    if k == 2:
        rz_h = [rng.integers(0, 5, N).astype(np.int32), rng.integers(0, 3, N).astype(np.int32)]     # operand 0 raised too
    rz_d = [None if r_ is None else DevArray(r_) for r_ in rz_h]
    want_r = [1] * N
    for j, v in enumerate(vals):
        want_r = [a * pow(b, 1 << (0 if rz_h[j] is None else int(rz_h[j][i])), key.nsq) % key.nsq for i, (a, b) in enumerate(zip(want_r, v))]
    _native.check(_addn_call(nk, ops, rz_d, 0, 0, 0, N, out))
    assert limbs_to_ints(out.get()) == want_r
    _native.check(_addn_call(nk, ops, rz_d, 0, 0, 1 - k, N, out))                    # raised tiles are brought to the same tag
    assert limbs_to_ints(out.get()) == [w * rpow(1 - k) % key.nsq for w in want_r]
    # in place: the output aliases operand 1
    _native.check(_addn_call(nk, ops, None, 0, 0, 0, N, ops[1]))
    assert limbs_to_ints(ops[1].get()) == want
    # argument checks
    assert _addn_call(nk, ops[:1], None, 0, 0, 0, N, out) == _native.PAI_E_INVALID
    assert _addn_call(nk, ops, None, 0, 0, 500, N, out) == _native.PAI_E_INVALID


def test_many_keys_take_the_small_table_and_do_not_thrash(monkeypatch):
    """A process that holds many DJN keys on one device (a federated server: one key per party): the first keys get the big
    fixed-base table, later ones the small operating point, under a 16 GB cache budget nothing is evicted and rebuilt on the
    second round of encryptions, and every key's ciphertexts are the oracle's — VERDICT r04 #6 / ADVICE r04 (cache accounting
    from the real allocation sizes)."""
    import torch

    from pailliercryptolib_python_amd import engine

    monkeypatch.setenv("PAI_FB_CACHE_MB", "16384")
    monkeypatch.setenv("PAI_LATENCY_MAX", "0")                      # the fixed-base tables for every batch size
    base = bench_key()
    NK, N = 32, 64
    keys = [orc.make_key(orc.BENCH_P, orc.BENCH_Q, djn_x=0x1000 + 7 * i, bits=2048) for i in range(NK)]
    pubs = [engine.PublicKeyHandle(k.n, 2048, k.hs, k.randbits, device="cuda:0") for k in keys]
    m = plaintexts(base, N, 5)
    r_l = orc.synth_r_limbs(9, N, base.randbits)
    dm = engine.to_device_words(ints_to_limbs(m, pubs[0].n_words), pubs[0].device)
    dr = engine.to_device_words(r_l, pubs[0].device)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    first = [engine.to_host_words(p.encrypt(dm, dr)) for p in pubs]
    info = [p.table_info() for p in pubs]
    assert all(i["bytes"] > 0 for i in info), "a table was evicted during the first round"
    big = [i for i in info if i["bytes"] > (1 << 30)]
    small = [i for i in info if i["bytes"] <= (256 << 20)]
    assert 1 <= len(big) <= 8 and len(big) + len(small) == NK, [i["bytes"] >> 20 for i in info]
    assert sum(i["bytes"] for i in info) <= 16384 << 20
    used = free0 - torch.cuda.mem_get_info()[0]
    assert used <= (20 << 30), used                                  # tables + per-key scratch stay near the budget
    second = [engine.to_host_words(p.encrypt(dm, dr)) for p in pubs]
    assert [p.table_info() for p in pubs] == info, "tables were rebuilt between the rounds"
    for k, a, b in zip(keys, first, second):
        assert np.array_equal(a, b)
    r_int = orc.limbs_to_ints(r_l)
    for idx in (0, 1, 9, NK - 1):                                   # a big-table key and small-table keys against the oracle
        assert limbs_to_ints(first[idx][:4]) == [orc.encrypt(keys[idx], x, rr) for x, rr in zip(m[:4], r_int[:4])]


def test_host_stage_operands_are_read_in_place(k2048):
    """pai_host_stage (include/paillier_hip.h): exponents and shifts staged in the pinned ring are read by the kernels through
    the returned pointers — same results as device-resident operands — over more stagings than the ring has slots, two parts
    in one slot, and the size limit."""
    key, N = k2048.key, 24
    rng = np.random.default_rng(4242)
    a = rand_below(rng, key.nsq, N)
    b = rand_below(rng, key.nsq, N)
    da, db = DevArray(ints_to_limbs(a, k2048.cw)), DevArray(ints_to_limbs(b, k2048.cw))
    outs, wants = [], []
    for it in range(40):                                     # 40 stagings: the 32-slot ring wraps, no synchronisation in between
        es = [int(v) | 1 << 52 for v in rng.integers(0, 1 << 52, N)]
        e_h = ints_to_limbs(es, 2)
        delta = rng.integers(-3, 4, N).astype(np.int32)
        srcs = (C.c_void_p * 2)(e_h.ctypes.data, delta.ctypes.data)
        sizes = (C.c_size_t * 2)(e_h.nbytes, delta.nbytes)
        ptrs = (C.c_void_p * 2)()
        _native.check(k2048.lib.pai_host_stage(0, 2, srcs, sizes, None, ptrs))
        assert ptrs[0] and ptrs[1] and ptrs[1] - ptrs[0] == (e_h.nbytes + 15) // 16 * 16
        o1, o2 = DevArray(shape=(N, k2048.cw)), DevArray(shape=(N, k2048.cw))
        _native.check(k2048.lib.pai_ct_mul(k2048.pk, da.ptr, C.c_void_p(ptrs[0]), 2, 53, 0, N, o1.ptr, None))
        _native.check(k2048.lib.pai_ct_add_aligned(k2048.pk, da.ptr, db.ptr, 0, C.c_void_p(ptrs[1]), N, o2.ptr, None))
        outs.append((o1, o2))
        wants.append((pow_many(a, es, key.nsq),
                      [pow(x, 1 << max(0, -int(d)), key.nsq) * pow(y, 1 << max(0, int(d)), key.nsq) % key.nsq for x, y, d in zip(a, b, delta)]))
    for (o1, o2), (w1, w2) in zip(outs, wants):
        assert limbs_to_ints(o1.get()) == w1 and limbs_to_ints(o2.get()) == w2
    # the calls that take the host operand themselves (stage + launch in one): pai_ct_mul_host, pai_ct_add_aligned_host
    es = [int(v) | 1 << 52 for v in rng.integers(0, 1 << 52, N)]
    e_h = ints_to_limbs(es, 2)
    delta = rng.integers(-3, 4, N).astype(np.int32)
    o1, o2 = DevArray(shape=(N, k2048.cw)), DevArray(shape=(N, k2048.cw))
    _native.check(k2048.lib.pai_ct_mul_host(k2048.pk, da.ptr, e_h.ctypes.data, 2, 53, 0, N, o1.ptr, None))
    _native.check(k2048.lib.pai_ct_add_aligned_host(k2048.pk, da.ptr, db.ptr, 0, delta.ctypes.data, N, o2.ptr, None))
    assert limbs_to_ints(o1.get()) == pow_many(a, es, key.nsq)
    assert limbs_to_ints(o2.get()) == [pow(x, 1 << max(0, -int(d)), key.nsq) * pow(y, 1 << max(0, int(d)), key.nsq) % key.nsq
                                       for x, y, d in zip(a, b, delta)]
    big_e = np.zeros((3000, 2), dtype=np.uint32)
    assert k2048.lib.pai_ct_mul_host(k2048.pk, da.ptr, big_e.ctypes.data, 2, 53, 0, 3000, o1.ptr, None) == -1      # more than a slot
    big = np.zeros(5000, dtype=np.uint8)
    srcs = (C.c_void_p * 1)(big.ctypes.data)
    sizes = (C.c_size_t * 1)(big.nbytes)
    ptrs = (C.c_void_p * 1)()
    assert k2048.lib.pai_host_stage(0, 1, srcs, sizes, None, ptrs) == -1


def _last_kernels(lib):
    names, i = [], 0
    name, ms = C.create_string_buffer(64), C.c_float(0)
    while lib.pai_profile_last(i, name, 64, C.byref(ms)) == 0:
        names.append(name.value.decode())
        i += 1
    return names


@pytest.mark.parametrize("bits", [1024, 2048, 3072, 4096])
def test_ct_add_by_one_msb_first_product(bits, monkeypatch):
    """pai_ct_add of wire-form batches beyond the small-batch range: ONE most-significant-limb-first product on lane groups
    (csrc/mont_msb.hpp, k_modmul_msb) instead of two Montgomery products — every element against CPython at the four key sizes'
    geometries (36 x 2, 36 x 4, 28 x 8, 36 x 8): residues that make the quotient digits, the top cells and the final subtraction
    extreme, operands that are NOT reduced (any word pattern of the row), ragged tiles, in place, and the same bits from the
    Montgomery route on a large batch.  classes.cpp:318-321."""
    K = NativeKey(seeded_key(bits))
    key, M, W = K.key, K.key.nsq, K.cw
    n = key.n
    full = (1 << (32 * W)) - 1
    rng = np.random.default_rng(bits + 1)
    special = [0, 1, 2, n - 1, n, n + 1, M - 1, M - 2, M - n, (n - 1) * n, M // 2, (1 << (M.bit_length() - 1)) - 1, 1 << (M.bit_length() - 1),
               full, full - 1, full >> 1, M, M + 1, 2 * M - 1 if 2 * M - 1 <= full else full, (1 << 29) - 1, (1 << (29 * 5)) - 1]
    monkeypatch.setenv("PAI_LAT_ADD_MAX", "0")              # the throughput route at every batch size
    _native.check(K.lib.pai_profile_enable(1))
    try:
        for N in (1, 33, 257, 700):
            a = (special + rand_below(rng, M, N))[:N] if N > 30 else rand_below(rng, M, N)
            b = (list(reversed(special)) + rand_below(rng, M, N))[:N] if N > 30 else rand_below(rng, M, N)
            if N == 700:
                a[40:61] = special                          # every special value against a rotation of the list
                b[40:61] = special[7:] + special[:7]
                a[100:110] = [int.from_bytes(rng.bytes(4 * W), "little") for _ in range(10)]      # unreduced words
                b[105:115] = [int.from_bytes(rng.bytes(4 * W), "little") for _ in range(10)]
                # the sum M - 1: before the final subtraction every lane above the first holds exactly the modulus' limbs, so the
                # borrow of the first lane runs through all of them (cond_sub's second and later rounds)
                a[120:124] = [M - 1, 1, n - 1, M - 1]
                b[120:124] = [1, M - 1, n + 1, M - 1]          # (n - 1)(n + 1) = M - 1;  (M - 1)^2 = 1
            da, db = DevArray(ints_to_limbs(a, W)), DevArray(ints_to_limbs(b, W))
            out = DevArray(shape=(N, W))
            _native.check(K.lib.pai_ct_add(K.pk, da.ptr, db.ptr, 0, N, out.ptr, None))
            assert _last_kernels(K.lib) == ["k_modmul_msb"]
            assert limbs_to_ints(out.get()) == [x * y % M for x, y in zip(a, b)], N
            _native.check(K.lib.pai_ct_add(K.pk, da.ptr, db.ptr, 0, N, da.ptr, None))         # in place
            assert limbs_to_ints(da.get()) == [x * y % M for x, y in zip(a, b)], N
    finally:
        _native.check(K.lib.pai_profile_enable(0))
    N = 20011
    a = rng.integers(0, 1 << 32, (N, W), dtype=np.uint64).astype(np.uint32)
    b = rng.integers(0, 1 << 32, (N, W), dtype=np.uint64).astype(np.uint32)
    a[::3, -1] &= 0x00FFFFFF                                 # two thirds unreduced
    da, db, o1, o2 = DevArray(a), DevArray(b), DevArray(shape=(N, W)), DevArray(shape=(N, W))
    _native.check(K.lib.pai_ct_add(K.pk, da.ptr, db.ptr, 0, N, o1.ptr, None))
    knob_disable(monkeypatch, "add_msb")
    _native.check(K.lib.pai_ct_add(K.pk, da.ptr, db.ptr, 0, N, o2.ptr, None))
    got, ref = o1.get(), o2.get()
    idx = [0, 1, 2, 255, 256, N // 2, N - 2, N - 1]
    assert limbs_to_ints(got[idx]) == [x * y % M for x, y in zip(limbs_to_ints(a[idx]), limbs_to_ints(b[idx]))]
    # (the Montgomery route is defined for reduced operands and agrees on every row that has them)
    red = np.arange(N) % 3 == 0
    ared = np.array([v < M for v in limbs_to_ints(a[red][:, :])])
    bred = np.array([v < M for v in limbs_to_ints(b[red][:, :])])
    rows = np.flatnonzero(red)[ared & bred]
    assert rows.size and np.array_equal(got[rows], ref[rows])


@pytest.mark.parametrize("lat_add", ["0", "4096"])
def test_ct_add_plain_is_the_product_with_the_raw_encryption(k2048, lat_add, monkeypatch):
    """pai_ct_add_plain (include/paillier_hip.h): ct * (1 + m n) mod n^2 in one pass — ipcl_python.py:495-504 followed by the
    ciphertext product of classes.cpp:318-321 — on the latency geometry and on lane groups, edge residues, in place."""
    monkeypatch.setenv("PAI_LAT_ADD_MAX", lat_add)
    key = k2048.key
    rng = np.random.default_rng(515)
    for N in (1, 16, 300):
        ct = rand_below(rng, key.nsq, N)
        m = rand_below(rng, key.n, N)
        m[0] = 0
        if N > 2:
            m[1], m[2], ct[2] = key.n - 1, 1, key.nsq - 1
        dct, dm, out = DevArray(ints_to_limbs(ct, k2048.cw)), DevArray(ints_to_limbs(m, k2048.nw)), DevArray(shape=(N, k2048.cw))
        _native.check(k2048.lib.pai_ct_add_plain(k2048.pk, dct.ptr, dm.ptr, N, out.ptr, None))
        want = [c * (1 + x * key.n) % key.nsq for c, x in zip(ct, m)]
        assert limbs_to_ints(out.get()) == want, (lat_add, N)
        _native.check(k2048.lib.pai_ct_add_plain(k2048.pk, dct.ptr, dm.ptr, N, dct.ptr, None))
        assert limbs_to_ints(dct.get()) == want, (lat_add, N, "in place")
