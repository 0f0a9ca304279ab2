"""GPU: key sizes at the engine-selection boundaries: the library picks a different kernel family per size —
digit-pair decrypt from 400-bit primes (24 / 36 / 56 / 72 limbs), wide engine or lane groups below, base-n digit
encryption for n of 700-1024 and 1400-2068 bits, lane groups elsewhere, host codec for n <= 66 bits — and
`generate_keypair` accepts any multiple of 4 from 64 bits.  Every size: ciphertext bits with explicit randomness,
decryption on both sides of the latency switch, add / mul / inverse, against the Python-int oracle."""
import numpy as np
import pytest

from oracle import paillier_oracle as orc
from pailliercryptolib_python_amd import PaillierPrivateKey, PaillierPublicKey
from pailliercryptolib_python_amd.bindings import ipclPublicKey

pytestmark = pytest.mark.gpu


def make(bits):
    if bits > 2560 and bits % 128 == 0:
        # (the oracle's CPython prime search takes seconds at these sizes: the native generator, seeded — its primes are checked
        # by tests/test_keygen_cpu.py)
        from pailliercryptolib_python_amd import _native

        p, q = _native.keygen(bits, True, seed=7000 + bits)
    else:
        p = orc.seeded_prime(bits // 2, 7000 + bits)
        q = orc.seeded_prime(bits // 2, 9000 + bits)
        while q == p or (p * q).bit_length() != bits:
            q = orc.seeded_prime(bits // 2, q % 100003)
    key = orc.make_key(p, q, djn_x=0xABCDEF1234567, bits=bits)
    pk = PaillierPublicKey(ipclPublicKey(key.n, bits, True, hs=key.hs, randbits=key.randbits))
    return key, pk, PaillierPrivateKey(pk, p, q)


@pytest.mark.parametrize("bits", [64, 128, 256, 512, 768, 800, 1280, 1536, 2560, 3328, 3584])
def test_key_size_boundaries(bits, monkeypatch):
    key, pk, sk = make(bits)
    rng = np.random.default_rng(bits)
    N = 37 if bits <= 2560 else 9                         # (the oracle's CPython pow dominates at the wide keys)
    small = bits <= 128                                   # max_int = n/3: keep |mantissa * 2^exponent| inside it
    vals = [int(v) for v in rng.integers(-1000, 1000, N)] if small else [float(v) for v in rng.uniform(-1000, 1000, N)]
    r = orc.synth_r_limbs(bits, N, key.randbits)
    en = pk.encrypt(vals, r=r)
    want_ct, want_e = orc.api_encrypt(key, vals, orc.limbs_to_ints(r))
    assert [int(c) for c in en.ciphertextBN()] == want_ct and en.exponent() == want_e
    for switch in ("0", "100000"):
        monkeypatch.setenv("PAI_LATENCY_MAX", switch)
        assert sk.decrypt(en) == orc.api_decrypt(key, want_ct, want_e) == vals
        assert sk.raw_decrypt(en) == [orc.decrypt_crt(key, c) for c in want_ct]
    monkeypatch.delenv("PAI_LATENCY_MAX")
    s = en + en
    want = orc.api_add_ct(key, want_ct, want_e, want_ct, want_e)
    assert ([int(c) for c in s.ciphertextBN()], s.exponent()) == (want[0], want[1])
    w = [int(v) for v in rng.integers(-9, 9, N)]         # negative multipliers: batch inversion
    pr = en * w
    want = orc.api_mul_plain(key, want_ct, want_e, w)
    assert ([int(c) for c in pr.ciphertextBN()], pr.exponent()) == (want[0], want[1])
    if bits >= 1024:
        # float multipliers (53-bit exponents): the small-batch ct * pt kernels — the four-wave pipeline where n k fits its rows
        # (one or two limbs per lane: 3328- and 3584-bit keys sit on either side of the 60-limb switch of the decryption chain)
        wf = [float(v) for v in rng.uniform(0.5, 3.0, N)]
        pr = en * wf
        want = orc.api_mul_plain(key, want_ct, want_e, wf)
        assert ([int(c) for c in pr.ciphertextBN()], pr.exponent()) == (want[0], want[1])
    tot = en.sum()
    want = orc.api_sum(key, want_ct, want_e)
    assert ([int(c) for c in tot.ciphertextBN()], tot.exponent()) == (want[0], want[1])
    if not small:
        assert abs(sk.decrypt(tot) - sum(vals)) < 1e-6


@pytest.mark.parametrize("bits,wbits", [(2304, "6"), (3072, "7"), (3072, None), (3200, "8"), (3264, "5"), (4096, "10"), (4128, "8"),
                                        (4160, "6")])
def test_lane_group_digit_pair_obfuscator(bits, wbits, monkeypatch):
    """DJN encryption and apply_obfuscator on lane-group digit pairs (kernels_pair.hpp: n of 2049..3228 bits on 112
    limbs, up to 4156 bits on 144 limbs; wider moduli keep the products modulo n^2) on the THROUGHPUT path (small
    batches would take the latency kernels): ciphertext bits against the oracle for one- and two-level table builds
    (odd / even window widths; None = the default 16 bits), and the same bits with the pair path switched off."""
    import ctypes as C

    from pailliercryptolib_python_amd import _native
    from tests._util import DevArray, djn_encrypt_many, djn_obfuscate_many, ints_to_limbs, limbs_to_ints, tune
    from tests._util import disable as knob_disable
    from tests.test_gpu_paillier_abi import NativeKey, plaintexts

    monkeypatch.setenv("PAI_LATENCY_MAX", "0")
    if wbits is not None:
        tune(monkeypatch, "fb_wbits", wbits)
    key, _, _ = make(bits)
    N = 70 if bits < 4000 else 37                         # a full and a ragged workgroup tile at either geometry (64 / 32 elements)
    m = plaintexts(key, N, bits)
    r = orc.synth_r_limbs(bits + 1, N, key.randbits)
    r[0] = 0
    r[1] = 0xFFFFFFFF
    if key.randbits % 32:
        r[1, -1] = (1 << (key.randbits % 32)) - 1
    r_int = orc.limbs_to_ints(r)
    want = djn_encrypt_many(key, m, r_int)                # the oracle's formula; bulk powers through the C oracle, spot-checked
    assert want[:2] == [orc.encrypt(key, x, rr) for x, rr in zip(m[:2], r_int[:2])]
    want2 = djn_obfuscate_many(key, want, r_int)
    for disable in ("0", "1"):
        knob_disable(monkeypatch, "pair", disable == "1")
        nk = NativeKey(key)
        dm, dr = DevArray(ints_to_limbs(m, nk.nw)), DevArray(r)
        ct = DevArray(shape=(N, nk.cw))
        _native.check(nk.lib.pai_encrypt(nk.pk, dm.ptr, dr.ptr, N, ct.ptr, None))
        assert limbs_to_ints(ct.get()) == want, (bits, wbits, disable)
        _native.check(nk.lib.pai_obfuscate(nk.pk, ct.ptr, dr.ptr, N, None))
        assert limbs_to_ints(ct.get()) == want2, (bits, wbits, disable)
        out = DevArray(shape=(N, nk.nw))
        _native.check(nk.lib.pai_decrypt(nk.sk, ct.ptr, N, out.ptr, None))
        assert limbs_to_ints(out.get()) == m
        del nk


@pytest.mark.parametrize("bits", [2304, 3072, 3200, 3264, 4096, 4128])
def test_lane_group_digit_pair_ct_times_pt(bits, monkeypatch):
    """ciphertext * plaintext for n of 2049 .. 4156 bits on lane-group digit pairs (kernels_pair.hpp: k_pair_ctmul — digit
    form through the base-R digits, per-slot power table, squarings at 4 NL^2 with the table entry streamed into LDS,
    k_pair_finish) on the THROUGHPUT path: bits against CPython pow for every window width the exponent widths select
    (2 / 3 / 4 / 5 bits), per-element and broadcast exponents, zero digits and zero exponents, a full and a ragged tile,
    in place — and the same bits with the pair path switched off (products modulo n^2)."""
    from pailliercryptolib_python_amd import _native
    from tests._util import DevArray, ints_to_limbs, limbs_to_ints, pow_many, rand_below
    from tests._util import disable as knob_disable
    from tests.test_gpu_paillier_abi import NativeKey

    monkeypatch.setenv("PAI_LATENCY_MAX", "0")
    key, _, _ = make(bits)
    M = key.nsq
    rng = np.random.default_rng(bits + 5)
    N = 70 if bits < 4000 else 37                         # a full and a ragged tile (64 / 32 elements per workgroup)
    c = rand_below(rng, M, N)
    c[0], c[1] = 1, M - 1
    for disable in ("0", "1"):
        knob_disable(monkeypatch, "pair_ctmul", disable == "1")
        nk = NativeKey(key)
        dc = DevArray(ints_to_limbs(c, nk.cw))
        for ebits in (12, 53, 130, 300):
            e = [int.from_bytes(rng.bytes(ebits // 8 + 1), "little") % (1 << ebits) for _ in range(N)]
            e[0], e[2], e[3] = 0, (1 << ebits) - 1, 1 << (ebits - 1)
            ew = (ebits + 31) // 32
            de = DevArray(ints_to_limbs(e, ew))
            out = DevArray(shape=(N, nk.cw))
            _native.check(nk.lib.pai_ct_mul(nk.pk, dc.ptr, de.ptr, ew, ebits, 0, N, out.ptr, None))
            assert limbs_to_ints(out.get()) == pow_many(c, e, M), (bits, ebits, disable)
            db = DevArray(ints_to_limbs([e[4]], ew))
            _native.check(nk.lib.pai_ct_mul(nk.pk, dc.ptr, db.ptr, ew, ebits, 1, N, out.ptr, None))
            assert limbs_to_ints(out.get()) == pow_many(c, e[4], M), (bits, ebits, disable, "bcast")
        d2 = DevArray(ints_to_limbs(c, nk.cw))
        e = [int(v) for v in rng.integers(1, 1 << 53, N)]
        de = DevArray(ints_to_limbs(e, 2))
        _native.check(nk.lib.pai_ct_mul(nk.pk, d2.ptr, de.ptr, 2, 53, 0, N, d2.ptr, None))                # in place
        assert limbs_to_ints(d2.get()) == pow_many(c, e, M), (bits, disable, "in place")
        del nk


@pytest.mark.parametrize("bits", [384, 800, 1280, 1536, 2560, 3584])
def test_wire_form_sum_on_lane_groups_at_odd_key_sizes(bits, monkeypatch):
    """pai_ct_add beyond the small-batch range at key sizes whose n^2 sits anywhere in its geometry: the most-significant-limb-first
    product (csrc/mont_msb.hpp) where the key has the context — 384 bits: one lane per integer; 800 bits: n^2 shifted up by the
    largest offset served (16 limbs) — and the two Montgomery products where it has not; reduced operands, every element against
    CPython, and the two routes against each other."""
    from pailliercryptolib_python_amd import engine
    from tests._util import disable as knob_disable

    key, pk, _ = make(bits)
    h = pk.pubkey.handle
    M = key.n * key.n
    rng = np.random.default_rng(bits + 5)
    N = 257
    W = h.ct_words
    a = [M - 1, 1, 0, M - 2] + [int.from_bytes(rng.bytes(4 * W), "little") % M for _ in range(N - 4)]
    b = [M - 1, M - 1, 7, 2] + [int.from_bytes(rng.bytes(4 * W), "little") % M for _ in range(N - 4)]
    ta = engine.to_device_words(engine.ints_to_words(a, W), h.device)
    tb = engine.to_device_words(engine.ints_to_words(b, W), h.device)
    monkeypatch.setenv("PAI_LAT_ADD_MAX", "0")
    got = engine.words_to_ints(engine.to_host_words(h.ct_add(ta, tb)))
    assert got == [x * y % M for x, y in zip(a, b)]
    knob_disable(monkeypatch, "add_msb")
    assert engine.words_to_ints(engine.to_host_words(h.ct_add(ta, tb))) == got
    # ... and on one integer per wavefront (16 / 32 / 64 lanes per integer: the borrow of (1, M - 1) -> M - 1 runs through every lane)
    monkeypatch.delenv("PAI_LAT_ADD_MAX")
    assert engine.words_to_ints(engine.to_host_words(h.ct_add(ta, tb))) == got
