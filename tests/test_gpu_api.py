"""GPU: the public API (PaillierKeypair / PaillierPublicKey / PaillierPrivateKey / PaillierEncryptedNumber).

(1) The reference's own tests (tests/ipcl_python_test.py) restated — with the assertions that file
    forgets for the matmul cases; (2) bit parity of every derived operation against the oracle's
    API-level compositions with the obfuscator randomness injected; (3) container / pickle behaviour
    exercised by example/ipclpy_example.py."""
import pickle

import numpy as np
import pytest

from oracle import paillier_oracle as orc
from pailliercryptolib_python_amd import (
    PaillierEncryptedNumber,
    PaillierKeypair,
    PaillierPrivateKey,
    PaillierPublicKey,
    context,
    engine,
    hybridControl,
    hybridMode,
)
from pailliercryptolib_python_amd import bindings as bindings_mod
from pailliercryptolib_python_amd.bindings import ipclPublicKey
from tests._util import tune

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def keys():
    return PaillierKeypair.generate_keypair(2048)


@pytest.fixture(scope="module")
def fixed():
    """Bench key with a known DJN base, and the matching oracle key."""
    okey = orc.make_key(orc.BENCH_P, orc.BENCH_Q, djn_x=0x1234567, bits=2048)
    raw = ipclPublicKey(okey.n, 2048, True, hs=okey.hs, randbits=okey.randbits)
    pk = PaillierPublicKey(raw)
    sk = PaillierPrivateKey(pk, orc.BENCH_P, orc.BENCH_Q)
    return pk, sk, okey


def ct_ints(enc: PaillierEncryptedNumber):
    return [int(b) for b in enc.ciphertextBN()]


# ---- (1) reference tests ---------------------------------------------------------------------------
def test_add_like_reference(keys):
    pk, sk = keys
    x_li = np.ones(100) * np.random.randint(100)
    y_li = np.ones(100) * np.random.randint(1000)
    z_li = np.ones(100) * np.random.rand()
    t_li = list(range(100))
    en_res = pk.encrypt(x_li) + pk.encrypt(y_li) + pk.encrypt(z_li) + pk.encrypt(t_li)
    res = x_li + y_li + z_li + t_li
    de = sk.decrypt(en_res)
    for i in range(100):
        assert round(abs(de[i] - res[i]), 7) == 0


def test_mul_like_reference(keys):
    pk, sk = keys
    x_li = np.ones(100) * np.random.randint(100)
    y_li = np.ones(100) * np.random.randint(1000) * -1
    z_li = np.ones(100) * np.random.rand()
    t_li = list(range(100))
    en_res = (pk.encrypt(x_li) * y_li + z_li) * t_li
    de = sk.decrypt(en_res)
    res = (x_li * y_li + z_li) * t_li
    for i in range(100):
        assert round(abs(de[i] - res[i]), 7) == 0
    x = 9
    en_x = pk.encrypt(x)
    for _ in range(12):
        en_x = en_x + 5000
        en_x = en_x - 0.2
        x = x + 5000 - 0.2
        assert round(abs(sk.decrypt(en_x) - x), 7) == 0


def test_matmul_family_with_real_assertions(keys):
    pk, sk = keys
    rng = np.random.default_rng(3)
    for _ in range(4):
        m, n, k = (int(v) for v in rng.integers(1, 7, 3))
        x, y = rng.random((m, n)), rng.random((n, k))
        en = pk.encrypt(x.flatten())
        got = np.array(sk.decrypt(en @ y)).reshape(m, k)
        assert np.allclose(got, x @ y)
        xl = rng.random((m, n)).tolist()
        en_y = pk.encrypt(y.flatten())
        got = np.array(sk.decrypt(xl @ en_y)).reshape(m, k)
        assert np.allclose(got, np.array(xl) @ y)
        en @= y
        assert np.allclose(np.array(sk.decrypt(en)).reshape(m, k), x @ y)
    v = rng.random(5)
    en = pk.encrypt(rng.random(15))
    assert len(en @ v) == 3
    with pytest.raises(ValueError):
        pk.encrypt(rng.random(7)) @ rng.random((3, 2))


def test_sum_mean_dot(keys):
    pk, sk = keys
    x = np.array([1.5, -2.25, 3.0, 100.125, 7.0])
    en = pk.encrypt(x)
    assert abs(sk.decrypt(en.sum()) - x.sum()) < 1e-9
    assert abs(sk.decrypt(en.mean()) - x.mean()) < 1e-9
    w = [2.0, -1.0, 0.5, 3.0, -4.0]
    assert abs(sk.decrypt(en.dot(w)) - float(np.dot(x, w))) < 1e-9
    mixed = pk.encrypt([1, 2.5, 3])
    assert abs(sk.decrypt(mixed.sum()) - 6.5) < 1e-12


# ---- (2) bit parity of the compositions -------------------------------------------------------------
def test_encrypt_bits_with_injected_randomness(fixed):
    pk, sk, okey = fixed
    vals = [0.0, 1.0, -1.0, 1234.5678, -5111.2834, 7, -7, 10**30, 1e-5, 2.0**60]
    r_l = orc.synth_r_limbs(21, len(vals), okey.randbits)
    en = pk.encrypt(vals, r=r_l)
    want_ct, want_e = orc.api_encrypt(okey, vals, orc.limbs_to_ints(r_l))
    assert ct_ints(en) == want_ct and en.exponent() == want_e
    raw = pk.raw_encrypt(vals)
    assert ct_ints(raw) == orc.api_encrypt(okey, vals, None)[0]
    dec = sk.decrypt(en)
    odec = orc.api_decrypt(okey, want_ct, want_e)
    assert dec == odec and [type(a) for a in dec] == [type(b) for b in odec]   # int when exponent <= 0, as upstream
    assert sk.raw_decrypt(en) == [orc.decrypt_crt(okey, c) for c in want_ct]
    assert np.array_equal(sk.decrypt_to_numpy(en), np.array([float(v) for v in vals]))


def test_add_mul_sub_div_bits(fixed):
    pk, sk, okey = fixed
    rng = np.random.default_rng(5)
    N = 24
    a = rng.uniform(-1000, 1000, N)
    b = list(rng.integers(-50, 50, N).astype(int))          # ints: exponent 0 => alignment in both directions
    ra, rb = orc.synth_r_limbs(31, N, okey.randbits), orc.synth_r_limbs(32, N, okey.randbits)
    ea, eb = pk.encrypt(a, r=ra), pk.encrypt([int(v) for v in b], r=rb)
    oa = orc.api_encrypt(okey, a, orc.limbs_to_ints(ra))
    ob = orc.api_encrypt(okey, [int(v) for v in b], orc.limbs_to_ints(rb))
    # ct + ct (mixed exponents), ct + array, ct + scalar, scalar + ct
    s = ea + eb
    want = orc.api_add_ct(okey, *oa, *ob)
    assert (ct_ints(s), s.exponent()) == (want[0], want[1])
    c = rng.uniform(-10, 10, N)
    s2 = ea + c
    want = orc.api_add_plain(okey, *oa, c)
    assert (ct_ints(s2), s2.exponent()) == (want[0], want[1])
    s3 = 5000 + eb
    want = orc.api_add_plain(okey, *ob, 5000)
    assert (ct_ints(s3), s3.exponent()) == (want[0], want[1])
    # ct * vector with negatives (inversion path), ct * scalar, negative scalar
    p = ea * c
    want = orc.api_mul_plain(okey, *oa, c)
    assert (ct_ints(p), p.exponent()) == (want[0], want[1])
    p2 = eb * -3
    want = orc.api_mul_plain(okey, *ob, -3)
    assert (ct_ints(p2), p2.exponent()) == (want[0], want[1])
    p3 = 2.5 * ea
    want = orc.api_mul_plain(okey, *oa, 2.5)
    assert (ct_ints(p3), p3.exponent()) == (want[0], want[1])
    # subtraction: ct - ct, ct - array, array - ct ; division by a scalar
    d = ea - eb
    want = orc.api_sub_ct(okey, *oa, *ob)
    assert (ct_ints(d), d.exponent()) == (want[0], want[1])
    d2 = ea - list(c)
    want = orc.api_sub_plain(okey, *oa, list(c))
    assert (ct_ints(d2), d2.exponent()) == (want[0], want[1])
    d3 = c - ea
    neg = orc.api_mul_plain(okey, *oa, -1.0)
    want = orc.api_add_plain(okey, *neg, c)
    assert (ct_ints(d3), d3.exponent()) == (want[0], want[1])
    q = ea / 4.0
    want = orc.api_mul_plain(okey, *oa, 0.25)
    assert (ct_ints(q), q.exponent()) == (want[0], want[1])
    # and the values are right
    assert np.allclose(sk.decrypt(d), a - np.array(b))
    assert np.allclose(sk.decrypt(p), a * c)


@pytest.mark.parametrize("seed", range(6))
def test_random_expression_chains_match_the_oracle_bit_for_bit(fixed, seed):
    """Differential run: random chains of + - * / between ciphertext vectors, float / int vectors and scalars of very
    different magnitudes (so that exponents diverge and alignment runs in both directions), every intermediate
    ciphertext compared with the oracle's composition of the reference's rules, every decryption with numpy."""
    pk, sk, okey = fixed
    rng = np.random.default_rng(9000 + seed)
    N = int(rng.integers(1, 9))

    def rand_plain():
        kind = rng.integers(0, 4)
        if kind == 0:
            return rng.uniform(-1e3, 1e3, N)
        if kind == 1:
            return np.ldexp(rng.uniform(-1, 1, N), rng.integers(-40, 40, N))
        if kind == 2:
            return [int(v) for v in rng.integers(-10**6, 10**6, N)]
        return float(rng.uniform(-50, 50))

    def as_np(v):
        return np.asarray(v, dtype=np.float64)

    r0 = orc.synth_r_limbs(100 + seed, N, okey.randbits)
    x = rand_plain()
    if np.isscalar(x):
        x = rng.uniform(-5, 5, N)
    cur = pk.encrypt(x, r=r0)
    ocur = orc.api_encrypt(okey, list(x), orc.limbs_to_ints(r0))
    val = as_np(x)
    for step in range(6):
        op = rng.integers(0, 5)
        y = rand_plain()
        if op == 0:                                   # ct + plain
            cur, ocur, val = cur + y, orc.api_add_plain(okey, *ocur, y), val + as_np(y)
        elif op == 1:                                 # ct - plain
            cur, ocur, val = cur - y, orc.api_sub_plain(okey, *ocur, y), val - as_np(y)
        elif op == 2:                                 # ct * plain (negative multipliers take the inversion path)
            cur, ocur, val = cur * y, orc.api_mul_plain(okey, *ocur, y), val * as_np(y)
        elif op == 3:                                 # ct + ct with a fresh encryption of different exponents
            z = rand_plain()
            if np.isscalar(z):
                z = [z] * N
            rz = orc.synth_r_limbs(200 + 10 * seed + step, N, okey.randbits)
            ez, oz = pk.encrypt(z, r=rz), orc.api_encrypt(okey, list(z), orc.limbs_to_ints(rz))
            cur, ocur, val = cur + ez, orc.api_add_ct(okey, *ocur, *oz), val + as_np(z)
        else:                                         # ct - ct
            z = rng.uniform(-1, 1, N)
            rz = orc.synth_r_limbs(300 + 10 * seed + step, N, okey.randbits)
            ez, oz = pk.encrypt(z, r=rz), orc.api_encrypt(okey, list(z), orc.limbs_to_ints(rz))
            cur, ocur, val = cur - ez, orc.api_sub_ct(okey, *ocur, *oz), val - z
        assert (ct_ints(cur), list(cur.exponent()) if N > 1 else cur.exponent()) == \
               (ocur[0], list(ocur[1]) if N > 1 else ocur[1]), (seed, step, int(op))
    got = np.asarray(sk.decrypt(cur), dtype=np.float64).reshape(-1)
    assert np.allclose(got, val, rtol=1e-9, atol=1e-9 * max(1.0, float(np.abs(val).max())))


@pytest.mark.parametrize("mexp_min", [None, "1"])
def test_reductions_match_the_oracle_bit_for_bit(fixed, mexp_min, monkeypatch):
    """sum / mean / dot / @ / r@ / @= against the oracle's restatement of ipcl_python.py:746-930: the padded
    rotate-and-add tree (:810-827), the index maps (:777-808) and the per-row maximum-exponent alignment
    (:868-870) fix the ciphertext bits and the exponents; mixed int / float inputs make the exponents differ.
    mexp_min = "1" sends every float matrix product through pai_ct_multiexp (large products take it by default)."""
    if mexp_min is not None:
        monkeypatch.setenv("PAI_MEXP_MIN_TERMS", mexp_min)
        tune(monkeypatch, "mexp_lanes", 3)                        # chunks of several members on these small shapes
    pk, sk, okey = fixed
    rng = np.random.default_rng(77)
    for N in (1, 2, 5, 8, 13):
        vals = [float(v) for v in rng.uniform(-100, 100, N)]
        vals[0] = int(rng.integers(-9, 9))                               # exponent 0 next to ~46
        if N > 2:
            vals[2] = float(np.ldexp(rng.uniform(1, 2), -30))            # a much larger exponent
        r = orc.synth_r_limbs(500 + N, N, okey.randbits)
        en = pk.encrypt(vals, r=r)
        oc, oe = orc.api_encrypt(okey, vals, orc.limbs_to_ints(r))
        s = en.sum()
        want = orc.api_sum(okey, oc, oe)
        assert (ct_ints(s), s.exponent()) == (want[0], want[1]) and len(s) == 1
        mean = en.mean()
        want = orc.api_mean(okey, oc, oe)
        assert (ct_ints(mean), mean.exponent()) == (want[0], want[1])
        w = [float(v) for v in rng.uniform(-3, 3, N)]
        if N > 1:
            w[1] = 2                                                     # an int multiplier: plaintext exponent 0
        d = en.dot(w)
        want = orc.api_dot(okey, oc, oe, w)
        assert (ct_ints(d), d.exponent()) == (want[0], want[1])
        assert abs(sk.decrypt(s) - sum(vals)) < 1e-6 and abs(sk.decrypt(d) - float(np.dot(vals, w))) < 1e-6
    for (m, n, k) in ((1, 1, 1), (2, 3, 2), (3, 4, 1), (1, 5, 3), (4, 2, 5)):
        x = rng.uniform(-10, 10, (m, n))
        x[0, 0] = 3.0                                                    # still a float: exponents differ by magnitude
        y = rng.uniform(-5, 5, (n, k))
        y[n - 1, 0] = -2.0
        rx = orc.synth_r_limbs(600 + m * 49 + n * 7 + k, m * n, okey.randbits)
        en = pk.encrypt(x.flatten(), r=rx)
        oc, oe = orc.api_encrypt(okey, list(x.flatten()), orc.limbs_to_ints(rx))
        res = en @ y
        want = orc.api_matmul(okey, oc, oe, y)
        assert (ct_ints(res), res.exponent()) == (want[0], want[1]) and len(res) == m * k
        assert np.allclose(np.array(sk.decrypt(res)).reshape(m, k), x @ y)
        if k == 1:                                                       # 1-D right operand (ipcl_python.py:788-792)
            res1 = en @ y[:, 0]
            want1 = orc.api_matmul(okey, oc, oe, y[:, 0])
            assert (ct_ints(res1), res1.exponent()) == (want1[0], want1[1])
        ry = orc.synth_r_limbs(700 + m * 49 + n * 7 + k, n * k, okey.randbits)
        en_y = pk.encrypt(y.flatten(), r=ry)
        oyc, oye = orc.api_encrypt(okey, list(y.flatten()), orc.limbs_to_ints(ry))
        xl = x.tolist()
        rres = xl @ en_y
        want = orc.api_matmul(okey, oyc, oye, xl, rhs=True)
        assert (ct_ints(rres), rres.exponent()) == (want[0], want[1])
        assert np.allclose(np.array(sk.decrypt(rres)).reshape(m, k), x @ y)
        if m == 1:                                                       # 1-D left operand
            rres1 = x[0] @ en_y
            want1 = orc.api_matmul(okey, oyc, oye, x[0], rhs=True)
            assert (ct_ints(rres1), rres1.exponent()) == (want1[0], want1[1])
        en2 = pk.encrypt(x.flatten(), r=rx)
        en2 @= y
        assert ct_ints(en2) == orc.api_matmul(okey, oc, oe, y)[0]
        yi = rng.integers(-5, 6, (n, k))                                # integer matrix: plaintext exponents 0, zeros included
        resi = en @ yi
        wanti = orc.api_matmul(okey, oc, oe, yi)
        assert (ct_ints(resi), resi.exponent()) == (wanti[0], wanti[1])
        assert np.allclose(np.array(sk.decrypt(resi)).reshape(m, k), x @ yi)


@pytest.mark.parametrize("bits", [1024, 3072, 4096])
def test_matrix_product_routes_agree_at_other_key_sizes(bits, monkeypatch):
    """`@`, `r@` and dot through pai_ct_multiexp (base-n digit pairs up to 2048-bit keys, lane groups above) give the
    bits and exponents of the term-by-term route AND of the oracle's api_matmul / api_dot, and decrypt to the plain product."""
    from tests.test_gpu_paillier_abi import seeded_key
    okey = seeded_key(bits)
    pk = PaillierPublicKey(ipclPublicKey(okey.n, bits, True, hs=okey.hs, randbits=okey.randbits))
    sk = PaillierPrivateKey(pk, okey.p, okey.q)
    rng = np.random.default_rng(bits)
    m, n, k = 3, 7, 4
    x, y = rng.uniform(-10, 10, (m, n)), rng.standard_normal((n, k))
    y[2, 1] = 0.0
    rx, ry = orc.synth_r_limbs(bits + 1, m * n, okey.randbits), orc.synth_r_limbs(bits + 2, n * k, okey.randbits)
    en = pk.encrypt(x.flatten(), r=rx)
    en_y = pk.encrypt(y.flatten(), r=ry)
    v = rng.standard_normal(m * n)
    # the oracle's restatement of ipcl_python.py:829-880 / :767-775 at this key size (bits AND exponents)
    oc, oe = orc.api_encrypt(okey, list(x.flatten()), orc.limbs_to_ints(rx))
    oyc, oye = orc.api_encrypt(okey, list(y.flatten()), orc.limbs_to_ints(ry))
    want = [tuple(orc.api_matmul(okey, oc, oe, y)), tuple(orc.api_matmul(okey, oyc, oye, x, rhs=True)),
            tuple(orc.api_dot(okey, oc, oe, list(v)))]
    got = {}
    for route, env in (("multiexp", "1"), ("terms", str(1 << 60))):
        monkeypatch.setenv("PAI_MEXP_MIN_TERMS", env)
        tune(monkeypatch, "mexp_lanes", 5)
        a, b, c = en @ y, x @ en_y, en.dot(v)
        got[route] = [(ct_ints(t), list(np.atleast_1d(t.exponent()))) for t in (a, b, c)]
        assert np.allclose(np.array(sk.decrypt(a)).reshape(m, k), x @ y) and np.allclose(np.array(sk.decrypt(b)).reshape(m, k), x @ y)
        assert abs(sk.decrypt(c) - float(np.dot(x.flatten(), v))) < 1e-6
    assert got["multiexp"] == got["terms"]
    assert got["terms"] == [(w[0], list(w[1])) for w in want]


def test_broadcast_rules(fixed):
    pk, sk, okey = fixed
    vec = pk.encrypt([1.0, 2.0, 3.0], r=orc.synth_r_limbs(1, 3, okey.randbits))
    one = pk.encrypt(10, r=orc.synth_r_limbs(2, 1, okey.randbits))
    assert sk.decrypt(vec + one) == [11.0, 12.0, 13.0]
    assert sk.decrypt(one + vec) == [11.0, 12.0, 13.0]            # length-1 left operand swaps (ipcl_python.py:369-375)
    two = pk.encrypt([1.0, 2.0])
    with pytest.raises(ValueError):
        vec + two
    with pytest.raises(ValueError):
        vec + [1.0, 2.0]
    with pytest.raises(ValueError):
        vec * [1.0, 2.0]
    other_pk, _ = PaillierKeypair.generate_keypair(1024)
    with pytest.raises(ValueError):
        vec + other_pk.encrypt([1.0, 2.0, 3.0])


# ---- (3) containers, errors, pickling ----------------------------------------------------------------
def test_container_protocol(fixed):
    pk, sk, okey = fixed
    en = pk.encrypt(list(range(10)))
    assert len(en) == en.length() == 10
    assert sk.decrypt(en[3]) == 3 and sk.decrypt(en[2:5]) == [2, 3, 4]
    assert [sk.decrypt(e) for e in en] == list(range(10))
    assert en.exponent(4) == 0 and int(en.ciphertextBN(4)) == ct_ints(en)[4]
    for bad in (10, -1):
        with pytest.raises(IndexError):
            en[bad]
        with pytest.raises(IndexError):
            en.exponent(bad)
        with pytest.raises(IndexError):
            en.ciphertextBN(bad)
    before = ct_ints(en)
    en.apply_obfuscator()
    assert ct_ints(en) != before and sk.decrypt(en) == list(range(10))
    with pytest.raises(ValueError):
        pk.encrypt(np.ones((2, 2)))
    with pytest.raises(ValueError):
        pk.encrypt(["a"])
    with pytest.raises(TypeError):
        pk.encrypt(np.array([1, 2], dtype=np.uint8))
    wrong_pk, wrong_sk = PaillierKeypair.generate_keypair(1024)
    with pytest.raises(ValueError):
        wrong_sk.decrypt(en)


def test_pickle_roundtrip_like_the_example(fixed):
    pk, sk, okey = fixed
    en = pk.encrypt([1.5, -2.5, 3.0])
    pk2 = pickle.loads(pickle.dumps(pk))
    sk2 = pickle.loads(pickle.dumps(sk))
    en2 = pickle.loads(pickle.dumps(en))
    assert pk2 == pk and sk2 == sk and pk2.n == pk.n
    assert ct_ints(en2) == ct_ints(en) and en2.exponent() == en.exponent()
    assert sk2.decrypt(en2) == [1.5, -2.5, 3.0]
    assert sk.decrypt(pk2.encrypt(4.25) + en2[0]) == 5.75
    state = pk.pubkey.__getstate__()
    assert state[0] == 1 and state[2] == 2048 and state[4] == 1024 and len(state[1]) == 256


def test_copy_constructor_and_qat_shims(fixed):
    pk, sk, okey = fixed
    assert PaillierPublicKey(pk).n == pk.n
    assert context.initializeContext("QAT") is True and context.isQATRunning() is False
    hybridControl.setHybridMode(hybridMode.OPTIMAL)
    assert hybridControl.getHybridMode() == hybridMode.OPTIMAL
    hybridControl.setHybridOff()
    assert context.terminateContext() is True


def test_key_sizes_1024_default_and_non_djn():
    pk, sk = PaillierKeypair.generate_keypair()           # n_length=1024, DJN
    x = np.random.default_rng(9).uniform(-1e6, 1e6, 100)  # BASELINE config[0]: 100 random floats, bit-exact round trip
    assert np.array_equal(np.array(sk.decrypt(pk.encrypt(x))), x)
    pk2, sk2 = PaillierKeypair.generate_keypair(1024, False)
    assert sk2.decrypt(pk2.encrypt([1.25, -7]) * 2) == [2.5, -14]


# ---- round 2: single-process multi-device fan-out, handle cache --------------------------------------------------
def test_fanout_over_a_device_list_gives_the_same_bits(fixed, monkeypatch):
    """The key replicated on a device list (here cuda:0 twice — the sharding, the per-device threads, the peer
    scatter / gather and the host-side concatenation are the code that runs over 8 GPUs): encrypt, ct*pt and decrypt
    of a batch large enough to shard equal the single-device results bit for bit."""
    import torch
    from pailliercryptolib_python_amd import bindings

    pk, sk, okey = fixed
    monkeypatch.setattr(bindings, "FANOUT_MIN_PER_DEVICE", 64)
    N = 64 * 3 + 17
    rng = np.random.default_rng(123)
    x = rng.uniform(-1000, 1000, N)
    r = orc.synth_r_limbs(77, N, okey.randbits)
    single = pk.encrypt(x, r=r)
    raw2 = ipclPublicKey(okey.n, 2048, True, hs=okey.hs, randbits=okey.randbits, devices=["cuda:0", "cuda:0", "cuda:0"])
    pk2 = PaillierPublicKey(raw2)
    sk2 = PaillierPrivateKey(pk2, orc.BENCH_P, orc.BENCH_Q)
    assert raw2.fanout_devices(N) is not None and raw2.fanout_devices(100) is None
    multi = pk2.encrypt(x, r=r)
    assert torch.equal(multi.words, single.words) and multi.exponent() == single.exponent()
    assert torch.equal(pk2.raw_encrypt(x).words, pk.raw_encrypt(x).words)
    xi = rng.integers(-10**9, 10**9, N)
    assert torch.equal(pk2.encrypt(xi, r=r).words, pk.encrypt(xi, r=r).words)
    w = rng.uniform(-3, 3, N)
    assert torch.equal((multi * w).words, (single * w).words)
    assert np.array_equal(sk2.decrypt_to_numpy(multi), x) and sk2.decrypt(multi) == sk.decrypt(single)
    assert sk2.raw_decrypt(multi) == sk.raw_decrypt(single)
    fresh = pk2.encrypt(x)                                     # device-drawn randomness per shard
    assert np.array_equal(sk2.decrypt_to_numpy(fresh), x)
    assert raw2.handle is pk.pubkey.handle                     # same key material + device => one cached handle


def test_unpickled_ciphertexts_share_one_handle_and_stay_on_the_host_until_used(fixed):
    pk, sk, okey = fixed
    en = pk.encrypt([1.5, -2.0, 3.25], r=orc.synth_r_limbs(9, 3, okey.randbits))
    blobs = [pickle.dumps(en) for _ in range(4)]
    back = [pickle.loads(b) for b in blobs]
    for b in back:
        assert b.ciphertext()._dev is None                     # nothing uploaded, no handle touched
        assert ct_ints(b) == ct_ints(en)
    s = back[0] + back[1]
    assert sk.decrypt(s) == [3.0, -4.0, 6.5]
    assert back[2].public_key.pubkey.handle is pk.pubkey.handle


def test_obfuscator_pool_encryptions_are_valid_and_consume_the_pool(fixed):
    """pooled obfuscators.  A pooled encryption is raw_encrypt(m) * obf with obf = E(0; r): it decrypts
    to m, differs from the raw ciphertext, equals the oracle's product of the two, and every pooled value is used once."""
    import torch

    pk, sk, okey = fixed
    pub = pk.pubkey
    pub._obf_pool = None
    pk.precompute_obfuscators(40)
    assert pub.obfuscator_pool_size() == 40
    pool = [int(v) for v in engine.words_to_ints(engine.to_host_words(pub._obf_pool))]
    assert len(set(pool)) == 40 and all(orc.decrypt_crt(okey, c) == 0 for c in pool[:5])       # encryptions of zero
    x = [1.5, -2.25, 1000.0, 7]
    en = pk.encrypt(x)
    assert pub.obfuscator_pool_size() == 36
    raw_ct, raw_e = orc.api_encrypt(okey, x, None)
    assert ct_ints(en) == [orc.ct_add(a, b, okey.nsq) for a, b in zip(raw_ct, pool[:4])] and en.exponent() == raw_e
    assert sk.decrypt(en) == [1.5, -2.25, 1000.0, 7]
    en2 = pk.encrypt(list(range(36)))
    assert pub.obfuscator_pool_size() == 0 and sk.decrypt(en2) == list(range(36))
    en3 = pk.encrypt([3.0, 4.0])                               # pool empty: direct encryption
    assert sk.decrypt(en3) == [3.0, 4.0]
    assert torch.equal(pk.raw_encrypt([5]).words, pk.raw_encrypt([5]).words)


def test_standard_scheme_randomness_is_drawn_on_the_device_inside_1_n():
    """Non-DJN keys: r uniform in [1, n) by rejection sampling on the device (ChaCha20 candidates of bits(n) bits)."""
    pk, sk = PaillierKeypair.generate_keypair(1024, False)
    n = pk.n
    r = pk.pubkey._draw_r(5000)
    vals = engine.words_to_ints(engine.to_host_words(r))
    assert all(0 < v < n for v in vals) and len(set(vals)) == 5000
    assert max(vals).bit_length() == n.bit_length()                       # candidates span the whole range
    x = np.random.default_rng(3).uniform(-5, 5, 300)
    en = pk.encrypt(x)
    assert np.array_equal(sk.decrypt_to_numpy(en), x)
    assert sk.decrypt(en + en) == [2 * float(v) for v in x]
    # a modulus just above a power of two rejects almost half of the candidates: still terminates, still inside [1, n)
    import torch
    from pailliercryptolib_python_amd.bindings import _rows_not_in_1_n

    n_small = (1 << 1023) + 12345
    n_w = torch.from_numpy(engine.int_to_words(n_small, 32).astype(np.int64)).to(r.device)
    bad = _rows_not_in_1_n(r, n_w)
    want = [not (0 < v < n_small) for v in vals]
    assert bad.cpu().tolist() == want and 0.05 < sum(want) / len(want) < 0.8       # n >= 0.5625 * 2^1024: at least 11 % lie above n_small


def test_standard_scheme_key_declared_wider_than_its_modulus():
    """ipclPublicKey(n, bits) only rejects n WIDER than bits: a modulus more than a 32-bit word shorter than the declared
    length must still draw its randomness inside [1, n) (the rows are key-length wide, the live top word is bits(n)'s)."""
    okey = orc.make_key(orc.seeded_prime(480, 91), orc.seeded_prime(480, 92))
    n = okey.n
    assert n.bit_length() <= 1024 - 32
    raw = ipclPublicKey(n, 1024, False)
    pk = PaillierPublicKey(raw)
    sk = PaillierPrivateKey(pk, okey.p, okey.q)
    r = raw._draw_r(3000)
    vals = engine.words_to_ints(engine.to_host_words(r))
    assert all(0 < v < n for v in vals) and len(set(vals)) == 3000 and max(vals).bit_length() == n.bit_length()
    x = np.random.default_rng(4).uniform(-50, 50, 200)
    en = pk.encrypt(x)
    assert np.array_equal(sk.decrypt_to_numpy(en), x)
    en.apply_obfuscator()
    assert np.array_equal(sk.decrypt_to_numpy(en), x)
    # the binding-level list forms (ipcl_bindings_classes.cpp:61-70,134-141)
    from pailliercryptolib_python_amd.bindings import ipclCipherText, ipclPlainText, ipclPrivateKey
    from pailliercryptolib_python_amd.paillier import BNUtils
    pt = ipclPlainText([BNUtils.int2BN(v) for v in (0, 1, 2, 12345, n - 1)])
    cts = raw.encrypt_tolist(pt, True)
    assert len(cts) == 5 and all(0 < int(c) < n * n for c in cts)
    back = ipclPrivateKey(raw, okey.p, okey.q).decrypt_tolist(ipclCipherText(raw, cts))
    assert [int(b) for b in back] == [0, 1, 2, 12345, n - 1]


def test_lazy_domain_tags_never_change_the_bits(fixed, monkeypatch):
    """(Small wire-form sums are exported at once since round 6 — bindings.EAGER_ADD_MAX; switched off here to reach the tags.)
    Sums are single Montgomery products whose stray R^-1 is remembered per container (bindings.ipclCipherText._raw);
    every boundary (ciphertextBN, pickling, decryption, slices, further mixed-exponent additions, products, the
    binding-level CipherText + CipherText and rotate) shows the bits of the reference's composition."""
    pk, sk, okey = fixed
    monkeypatch.setattr(bindings_mod, "EAGER_ADD_MAX", 0)
    rng = np.random.default_rng(17)
    N = 9
    vals = [np.round(rng.uniform(-100, 100, N), 3) for _ in range(4)]
    rs = [orc.synth_r_limbs(900 + i, N, okey.randbits) for i in range(4)]
    ens = [pk.encrypt(v, r=r) for v, r in zip(vals, rs)]
    ocs = [orc.api_encrypt(okey, list(v), orc.limbs_to_ints(r)) for v, r in zip(vals, rs)]
    # equal exponents inside each array pair? force it: integers encode with exponent 0
    ints = [list(range(i, i + N)) for i in (1, 50, 700)]
    ei = [pk.encrypt(v, r=orc.synth_r_limbs(950 + i, N, okey.randbits)) for i, v in enumerate(ints)]
    oi = [orc.api_encrypt(okey, v, orc.limbs_to_ints(orc.synth_r_limbs(950 + i, N, okey.randbits))) for i, v in enumerate(ints)]
    s2 = ei[0] + ei[1]
    s3 = s2 + ei[2]
    assert s2.ciphertext()._raw()[1] == -1 and s3.ciphertext()._raw()[1] == -2           # one product each, tags drift
    w2 = orc.api_add_ct(okey, *oi[0], *oi[1])
    w3 = orc.api_add_ct(okey, *w2, *oi[2])
    part = s3[2:5]                                                                        # a slice keeps the tag ...
    assert part.ciphertext()._raw()[1] == -2
    assert ct_ints(part) == w3[0][2:5]                                                    # ... and canonicalises on export
    assert ct_ints(s3) == w3[0] and s3.ciphertext()._raw()[1] == 0                       # exported once, cached in place
    assert sk.decrypt(s2) == [a + b for a, b in zip(ints[0], ints[1])]
    # mixed exponents on tagged operands: (float + float) + (int + int) aligns inside pai_ct_add_aligned_dom
    f2 = ens[0] + ens[1]
    mix = f2 + (ei[0] + ei[1])
    wf2 = orc.api_add_ct(okey, *ocs[0], *ocs[1])
    wmix = orc.api_add_ct(okey, *wf2, *w2)
    assert (ct_ints(mix), mix.exponent()) == (wmix[0], wmix[1])
    # tagged operand into a product, a difference, a reduction and a pickle
    s2b = ei[0] + ei[1]
    prod = s2b * 2.5
    wprod = orc.api_mul_plain(okey, *w2, 2.5)
    assert (ct_ints(prod), prod.exponent()) == (wprod[0], wprod[1])
    s2c = ei[0] + ei[1]
    diff = s2c - ei[2]
    wdiff = orc.api_sub_ct(okey, *w2, *oi[2])
    assert (ct_ints(diff), diff.exponent()) == (wdiff[0], wdiff[1])
    s2d = ei[0] + ei[1]
    tot = s2d.sum()
    wtot = orc.api_sum(okey, *w2)
    assert (ct_ints(tot), tot.exponent()) == (wtot[0], wtot[1])
    s2e = ei[0] + ei[1]
    back = pickle.loads(pickle.dumps(s2e))
    assert ct_ints(back) == w2[0]
    # binding level: CipherText + CipherText (classes.cpp:318-321), rotate keeps the tag, getTexts is the wire form
    ca, cb = ei[0].ciphertext(), ei[1].ciphertext()
    cs = ca + cb
    assert cs._raw()[1] == -1
    rot = cs.rotate(2)
    assert rot._raw()[1] == -1 and [int(b) for b in rot.getTexts()] == w2[0][2:] + w2[0][:2]
    assert [int(b) for b in cs.getTexts()] == w2[0]
    one = pk.encrypt(5, r=orc.synth_r_limbs(990, 1, okey.randbits))
    bs = ei[0] + one                                                                      # broadcast addend
    wb = orc.api_add_ct(okey, *oi[0], *orc.api_encrypt(okey, [5], orc.limbs_to_ints(orc.synth_r_limbs(990, 1, okey.randbits))))
    assert ct_ints(bs) == wb[0]


def test_key_trim_at_the_api_level(fixed):
    """ipclPublicKey.trim(): tables and scratch go, the next encryption rebuilds what it needs; results unchanged."""
    import torch

    pk, sk, okey = fixed
    x = np.random.default_rng(8).uniform(-9, 9, 5000)
    r = orc.synth_r_limbs(77, 5000, okey.randbits)
    a = pk.encrypt(x, r=r)
    first = a.words.clone()
    assert pk.pubkey.trim() > 0
    b = pk.encrypt(x, r=r)
    assert torch.equal(b.words, first)
    assert np.array_equal(sk.decrypt_to_numpy(b), x)


def test_large_mixed_exponent_sums_are_sorted_by_shift_and_bit_identical(fixed, monkeypatch):
    """Batches of >= 2^15 elements run the fused aligned addition in the order of |exponent difference| (paillier._add_aligned)
    and scatter the results back: the same ciphertext bits as the unsorted pass, a sample against the oracle's composition,
    and `a - b` through the same helper."""
    import torch

    pk, sk, okey = fixed
    rng = np.random.default_rng(99)
    N = (1 << 15) + 37
    x = rng.uniform(-1000, 1000, N)
    y = rng.uniform(-1000, 1000, N) * np.ldexp(1.0, rng.integers(-6, 7, N))
    ra, rb = orc.synth_r_limbs(1201, N, okey.randbits), orc.synth_r_limbs(1202, N, okey.randbits)
    a, b = pk.encrypt(x, r=ra), pk.encrypt(y, r=rb)
    got = {}
    for tag, env in (("sorted", "1"), ("plain", str(1 << 40))):
        monkeypatch.setenv("PAI_ALIGN_SORT_MIN", env)
        s, d = a + b, a - b
        got[tag] = (s.words.clone(), list(s.exponent()), d.words.clone(), list(d.exponent()))
    assert torch.equal(got["sorted"][0], got["plain"][0]) and got["sorted"][1] == got["plain"][1]
    assert torch.equal(got["sorted"][2], got["plain"][2]) and got["sorted"][3] == got["plain"][3]
    idx = [0, 1, 17, N // 2, N - 1]
    oa = orc.api_encrypt(okey, [float(x[i]) for i in idx], orc.limbs_to_ints(ra[idx]))
    ob = orc.api_encrypt(okey, [float(y[i]) for i in idx], orc.limbs_to_ints(rb[idx]))
    want = orc.api_add_ct(okey, *oa, *ob)
    sw = engine.words_to_ints(engine.to_host_words(got["sorted"][0][idx]))
    assert sw == want[0] and [got["sorted"][1][i] for i in idx] == want[1]
    monkeypatch.delenv("PAI_ALIGN_SORT_MIN")
    assert np.allclose(sk.decrypt_to_numpy(a + b), x + y, rtol=0, atol=1e-6 * np.abs(x + y).max())


def test_sums_of_sums_stay_lazy_and_tags_stay_bounded(fixed, monkeypatch):
    """(a+b)+(c+d) and x + acc keep their operands tagged (one product per addition: no operand is canonicalised just to
    read its shape), a long accumulator's tag is renormalised at paillier.DOM_MAX, the handle's R^k cache stays small
    and trim() drops it — bits always those of the reference's composition."""
    from pailliercryptolib_python_amd import paillier as P

    pk, sk, okey = fixed
    monkeypatch.setattr(bindings_mod, "EAGER_ADD_MAX", 0)         # (small batches would export each sum at once)
    N = 7
    ints = [list(range(i, i + N)) for i in (1, 50, 700, 9000)]
    rs = [orc.synth_r_limbs(970 + i, N, okey.randbits) for i in range(4)]
    e = [pk.encrypt(v, r=r) for v, r in zip(ints, rs)]
    o = [orc.api_encrypt(okey, v, orc.limbs_to_ints(r)) for v, r in zip(ints, rs)]
    ab, cd = e[0] + e[1], e[2] + e[3]
    tot = ab + cd
    assert ab.ciphertext()._raw()[1] == -1 and cd.ciphertext()._raw()[1] == -1         # still tagged after being operands
    assert tot.ciphertext()._raw()[1] == -3
    wab, wcd = orc.api_add_ct(okey, *o[0], *o[1]), orc.api_add_ct(okey, *o[2], *o[3])
    assert ct_ints(tot) == orc.api_add_ct(okey, *wab, *wcd)[0]
    right = e[0] + ab                                                                   # tagged RIGHT operand
    assert ab.ciphertext()._raw()[1] == -1 and right.ciphertext()._raw()[1] == -2
    assert ct_ints(right) == orc.api_add_ct(okey, *o[0], *wab)[0]
    acc, want = e[0], o[0]
    tags = []
    for k in range(2 * P.DOM_MAX + 3):
        acc = acc + e[1 + k % 3]
        want = orc.api_add_ct(okey, *want, *o[1 + k % 3])
        tags.append(acc.ciphertext()._raw()[1])
    assert min(tags) >= -P.DOM_MAX and 0 in tags[P.DOM_MAX:]                            # renormalised on the way
    assert ct_ints(acc) == want[0]
    h = pk.pubkey.handle
    assert len(h.__dict__.get("_dom_consts", {})) <= 32
    h.trim()
    assert "_dom_consts" not in h.__dict__
    assert ct_ints(e[0] + e[1]) == wab[0]


def test_plaintext_addends_are_aligned_in_the_plaintext_domain(fixed, monkeypatch):
    """ct + plaintext: the plaintext is encoded AT the ciphertext's exponent when its own is lower (pai_fp_encode_at /
    fixedpoint.align_encoded: (1 + m n)^(2^d) = 1 + (m 2^d mod n) n), so no ciphertext squaring runs for it — and the
    bits and exponents are exactly those of the reference's raw-encrypt-then-raise composition (oracle), for float
    arrays, int arrays, Python lists, scalars (broadcast) and the reference benchmark's BM_Add_CTPT shape."""
    pk, sk, okey = fixed
    h = pk.pubkey.handle
    calls = []
    real = type(h).ct_add_aligned
    monkeypatch.setattr(type(h), "ct_add_aligned", lambda self, *a, **k: calls.append(1) or real(self, *a, **k))
    N = 24
    ar = np.arange(N)
    x, y = (ar + 11) * 5111.2834, (32768 - ar) * 1.3872
    rx = orc.synth_r_limbs(31, N, okey.randbits)
    ex = pk.encrypt(x, r=rx)
    ox = orc.api_encrypt(okey, list(x), orc.limbs_to_ints(rx))
    exx, oxx = ex * x, orc.api_mul_plain(okey, *ox, list(x))                             # exponents ~70: far above y's ~37
    calls.clear()
    got = exx + y
    want = orc.api_add_plain(okey, *oxx, list(y))
    assert (ct_ints(got), got.exponent()) == (want[0], want[1])
    assert calls == []                                                                  # one plain product, no aligned-addition kernel
    for other in (np.arange(N, dtype=np.int64) - 5, [float(v) for v in y], [int(v) for v in range(N)], 3.75, -2, 0.0,
                  np.float64(1e-3), y * 2.0 ** 40, -y * 2.0 ** -30):
        got = exx + other
        want = orc.api_add_plain(okey, *oxx, other if np.isscalar(other) else list(other))
        assert (ct_ints(got), got.exponent()) == (want[0], want[1]), repr(other)[:40]
        got = other + ex                                                                 # __radd__, mixed directions
        want = orc.api_add_plain(okey, *ox, other if np.isscalar(other) else list(other))
        assert (ct_ints(got), got.exponent()) == (want[0], want[1]), repr(other)[:40]
    d = exx - y
    wd = orc.api_sub_plain(okey, *oxx, y)
    assert (ct_ints(d), d.exponent()) == (wd[0], wd[1])
    assert np.allclose(sk.decrypt_to_numpy(exx + y), x * x + y, rtol=1e-12)


def test_failed_inversion_is_raised_on_its_own_result(fixed):
    """a negative multiplier inverts the ciphertext asynchronously (pai_ct_invert_flag).  A non-invertible input
    must be reported on the ciphertext computed from it — on every export of it and of whatever was derived from it — and on
    nothing else: an unrelated ciphertext of the same key decrypts before and after, and the faulty one keeps failing (the
    outcome is not a per-handle word that the first reader clears)."""
    from pailliercryptolib_python_amd import _native
    from pailliercryptolib_python_amd.bindings import ipclCipherText

    pk, sk, okey = fixed
    good = pk.encrypt([1.5, -2.0, 3.25])
    bad_rows = [int(b) for b in good.ciphertextBN()]
    bad_rows[1] = okey.p * 977                                             # shares a factor with n: no inverse modulo n^2
    bad = PaillierEncryptedNumber(pk, ipclCipherText(pk.pubkey, bad_rows), good.exponent(), 3)
    res_bad = bad * -2.0                                                   # queued; nothing is read back here
    res_good = good * -2.0
    derived = res_bad + good                                               # the outcome travels with derived results
    sliced = res_bad[0:2]
    assert sk.decrypt(res_good) == [-3.0, 4.0, -6.5]                       # the unrelated result is not blamed ...
    for obj in (res_bad, derived, sliced, res_bad):                        # ... the faulty ones are, every time
        with pytest.raises(_native.NativeError, match="not invertible"):
            sk.decrypt(obj)
    with pytest.raises(_native.NativeError, match="not invertible"):
        res_bad.ciphertextBN()
    with pytest.raises(_native.NativeError, match="not invertible"):
        res_bad.ciphertextBN(0)
    with pytest.raises(_native.NativeError, match="not invertible"):
        _ = res_bad.words
    with pytest.raises(_native.NativeError, match="not invertible"):
        pickle.dumps(derived)
    with pytest.raises(_native.NativeError, match="not invertible"):
        sk.prikey.decrypt(res_bad.ciphertext())
    with pytest.raises(_native.NativeError, match="not invertible"):
        sk.decrypt(good - bad)                                             # ct - ct inverts the subtrahend
    assert sk.decrypt(good - good) == [0.0, 0.0, 0.0]
    assert sk.decrypt(res_good + good) == [-1.5, 2.0, -3.25]


def test_add_many_equals_the_chain_of_additions(fixed, monkeypatch):
    """PaillierEncryptedNumber.add_many (pai_ct_addn: the k-party aggregation in one pass) returns the ciphertext bits and
    exponents of the reference's chain a + b + c + ... (ipcl_python.py:365-381, 490-526), which the oracle restates:
    equal exponents, mixed exponents within the one-pass budget, lazily tagged operands, more than 16 operands, and the
    fallbacks (wide exponent spread, short arrays)."""
    pk, sk, okey = fixed
    monkeypatch.setattr(PaillierEncryptedNumber, "ADDN_MIN", 64)
    monkeypatch.setattr(bindings_mod, "EAGER_ADD_MAX", 0)         # lazily tagged operands at this test's small size
    rng = np.random.default_rng(77)
    N = 200

    def oracle_chain(arrays):
        ct, ex = orc.api_encrypt(okey, arrays[0], None)
        for a in arrays[1:]:
            c2, e2 = orc.api_encrypt(okey, a, None)
            ct, ex = orc.api_add_ct(okey, ct, ex, c2, e2)
        return ct, ex

    # equal exponents (integers), 5 operands
    arrs = [[int(v) for v in rng.integers(-1000, 1000, N)] for _ in range(5)]
    encs = [pk.raw_encrypt(a) for a in arrs]
    got = PaillierEncryptedNumber.add_many(encs)
    want_ct, want_e = oracle_chain(arrs)
    assert ct_ints(got) == want_ct and got.exponent() == want_e
    assert sk.decrypt(got) == [sum(col) for col in zip(*arrs)]
    # mixed exponents inside the budget: one operand one binade lower on part of the range
    f = [list(rng.uniform(512.0, 1023.0, N)) for _ in range(4)]
    f[2] = [v / 2 if i % 3 == 0 else v for i, v in enumerate(f[2])]
    encs = [pk.raw_encrypt(a) for a in f]
    got = PaillierEncryptedNumber.add_many(encs)
    want_ct, want_e = oracle_chain(f)
    assert ct_ints(got) == want_ct and got.exponent() == want_e
    # lazily tagged operands (sums of sums) and 19 operands (two chunks)
    many = [[int(v) for v in rng.integers(0, 50, N)] for _ in range(19)]
    encs = [pk.raw_encrypt(a) for a in many]
    encs[3] = encs[3] + pk.raw_encrypt([0] * N)                             # tag -1
    encs[0] = (encs[0] + pk.raw_encrypt([0] * N)) + pk.raw_encrypt([0] * N)     # tag -2 on the first operand
    got = PaillierEncryptedNumber.add_many(encs)
    assert sk.decrypt(got) == [sum(col) for col in zip(*many)]
    chain = encs[0]
    for e_ in encs[1:]:
        chain = chain + e_
    assert ct_ints(got) == ct_ints(chain) and got.exponent() == chain.exponent()
    # every operand a long lazy chain (tag -3 each, 35 of them: three chunks): the fix-up exponent of the natural tags would
    # leave the kernel's table of powers of R — add_many re-plans instead of raising
    deep = [[int(v) for v in rng.integers(0, 9, N)] for _ in range(35)]
    zero = pk.raw_encrypt([0] * N)
    encs = [((pk.raw_encrypt(a) + zero) + zero) + zero for a in deep]
    assert all(e_.ciphertext()._raw()[1] == -3 for e_ in encs)
    got = PaillierEncryptedNumber.add_many(encs)
    assert sk.decrypt(got) == [sum(col) for col in zip(*deep)]
    chain = encs[0]
    for e_ in encs[1:]:
        chain = chain + e_
    assert ct_ints(got) == ct_ints(chain) and got.exponent() == chain.exponent()
    # wide exponent spread: the chain itself (same bits by construction), still correct
    wide = [list(rng.uniform(-1000.0, 1000.0, N)) for _ in range(4)]
    encs = [pk.raw_encrypt(a) for a in wide]
    got = PaillierEncryptedNumber.add_many(encs)
    want_ct, want_e = oracle_chain(wide)
    assert ct_ints(got) == want_ct and got.exponent() == want_e


def test_small_host_operands_are_staged_and_never_change_the_bits(fixed, monkeypatch):
    """Round 6: shifts, exponents and codec inputs of small batches reach the kernels through a pinned ring (pai_host_stage)
    instead of a copy on the stream, the exponents of float batches are computed on the host, and small wire-form additions
    return the wire form at once.  (a) every operation gives the bits of the copy path (PAI_HOST_STAGE=0) and of the oracle;
    (b) 300 back-to-back calls without a synchronisation in between — the 32-slot ring wraps nine times while earlier kernels
    may still be queued — all decrypt correctly."""
    import torch

    pk, sk, okey = fixed
    rng = np.random.default_rng(2024)
    nb = 16
    x = (np.arange(nb) + 11) * 5111.2834
    y = (32768 - np.arange(nb)) * 1.3872
    r = engine.ints_to_words([int(v) for v in rng.integers(1, 1 << 62, nb)], pk.pubkey.handle.r_words)

    def run():
        cx = pk.encrypt(x, r=r)
        cy = pk.encrypt(y, r=r)
        return {"enc": (ct_ints(cx), cx.exponent()),
                "add": (ct_ints(cx + cy), (cx + cy).exponent()),
                "add_same": (ct_ints(cx + cx), (cx + cx).exponent()),
                "addpt": (ct_ints((cx * x) + y), ((cx * x) + y).exponent()),
                "mul": (ct_ints(cx * y), (cx * y).exponent()),
                "sub": (ct_ints(cx - cy), (cx - cy).exponent())}

    staged = run()
    monkeypatch.setenv("PAI_HOST_STAGE", "0")
    copied = run()
    monkeypatch.delenv("PAI_HOST_STAGE")
    assert staged == copied
    xc, xe = orc.api_encrypt(okey, list(x), engine.words_to_ints(r))
    yc, ye = orc.api_encrypt(okey, list(y), engine.words_to_ints(r))
    assert staged["enc"] == (xc, list(xe))
    ac, ae = orc.api_add_ct(okey, xc, xe, yc, ye)
    assert staged["add"] == (ac, list(ae))
    mc, me = orc.api_mul_plain(okey, xc, xe, list(y))
    assert staged["mul"] == (mc, list(me))
    assert (cx_tag := (pk.encrypt(x) + pk.encrypt(x)).ciphertext()._raw()[1]) == 0, cx_tag       # wire form at once
    # (b) a long unsynchronised run through the ring
    cx = pk.encrypt(x)
    outs, wants = [], []
    for i in range(300):
        yi = y + i
        outs.append((cx + yi) if i % 3 == 0 else ((cx * yi) if i % 3 == 1 else (cx + pk.encrypt(yi))))
        wants.append(x * yi if i % 3 == 1 else x + yi)
    torch.cuda.synchronize()
    for o, w in zip(outs, wants):
        assert np.allclose(sk.decrypt(o), w, rtol=1e-9, atol=1e-6)
