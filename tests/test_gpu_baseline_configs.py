"""GPU: the BASELINE.json configurations at their full sizes, checked through size-independent properties
(round trips, homomorphic identities evaluated with Python big ints on the residues) plus bit parity
against the C oracle on a sample.  One GPU runs one rank's shard of the 8-GPU configurations."""
import json
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import c_oracle as co
from oracle import paillier_oracle as orc
from pailliercryptolib_python_amd import PaillierKeypair, engine, fixedpoint, sharding

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def key2048():
    return orc.make_key(orc.BENCH_P, orc.BENCH_Q, djn_x=0x1234567, bits=2048)


def fixture_key(bits):
    fx = json.loads((Path(__file__).parent / "golden" / "fixture_keys.json").read_text())[str(bits)]
    return orc.make_key(int(fx["p"], 16), int(fx["q"], 16), djn_x=(1 << 70) + 12345, bits=bits)


def handles(key):
    pub = engine.PublicKeyHandle(key.n, key.bits, key.hs, key.randbits, device=DEV)
    return pub, engine.PrivateKeyHandle(pub, key.p, key.q)


def test_config0_1024bit_100_floats_roundtrip_bit_exact():
    pk, sk = PaillierKeypair.generate_keypair(1024)
    x = np.random.default_rng(1000).uniform(-1000, 1000, 100)
    assert np.array_equal(np.array(sk.decrypt(pk.encrypt(x))), x)
    # the same configuration on a fixed 1024-bit key with injected randomness: every ciphertext against the oracle
    from pailliercryptolib_python_amd import PaillierPrivateKey, PaillierPublicKey
    from pailliercryptolib_python_amd.bindings import ipclPublicKey

    key = fixture_key(1024)
    fpk = PaillierPublicKey(ipclPublicKey(key.n, 1024, True, hs=key.hs, randbits=key.randbits))
    fsk = PaillierPrivateKey(fpk, key.p, key.q)
    r_l = orc.synth_r_limbs(4000, 100, key.randbits)
    en = fpk.encrypt(x, r=r_l)
    want_ct, want_e = orc.api_encrypt(key, list(x), orc.limbs_to_ints(r_l))
    assert [int(c) for c in en.ciphertextBN()] == want_ct and en.exponent() == want_e
    assert fsk.decrypt(en) == orc.api_decrypt(key, want_ct, want_e) == [float(v) for v in x]


def test_config1_2048bit_batch_65536_encrypt_decrypt():
    key = key2048()
    pub, priv = handles(key)
    N = 65536
    x = np.random.default_rng(1001).uniform(-1000, 1000, N)
    res, expo = fixedpoint.encode_float64_array(x, key.n, pub.n_words)
    r_l = orc.synth_r_limbs(4001, N, key.randbits)
    m, r = engine.to_device_words(res, pub.device), engine.to_device_words(r_l, pub.device)
    ct = pub.encrypt(m, r)
    back = priv.decrypt(ct)
    torch.cuda.synchronize()
    assert torch.equal(back, m)                                              # every element round-trips
    assert np.array_equal(fixedpoint.decode_float64_array(engine.to_host_words(back), expo, key.n, key.max_int), x)
    # ciphertext bits against the C oracle on a 512-element sample spread over the batch
    idx = np.linspace(0, N - 1, 512).astype(np.int64)
    ck = co.COracleKey(key)
    want = ck.encrypt_djn(res[idx], r_l[idx])
    got = engine.to_host_words(ct[torch.from_numpy(idx).to(pub.device)])
    assert np.array_equal(got, want)
    assert np.array_equal(ck.decrypt_crt(want), res[idx])


def test_config2_2048bit_batch_1M_add_and_mul():
    key = key2048()
    pub, priv = handles(key)
    N = 1 << 20
    rng = np.random.default_rng(1002)
    a = rng.uniform(-1000, 1000, N)
    b = np.random.default_rng(3003).uniform(-1000, 1000, N)
    ra, ea = fixedpoint.encode_float64_array(a, key.n, pub.n_words)
    rb, eb = fixedpoint.encode_float64_array(b, key.n, pub.n_words)
    ca = pub.raw_encrypt(engine.to_device_words(ra, pub.device))
    cb = pub.raw_encrypt(engine.to_device_words(rb, pub.device))
    # ct + ct: D(E(a) E(b)) = a + b mod n  (same exponent classes only: compare residues, no alignment here)
    cs = pub.ct_add(ca, cb)                                # wire form in and out: one most-significant-limb-first product per element
    s = priv.decrypt(cs)
    # ... and the same bits, whole batch, from the lazy route (one Montgomery product, then the retag product at the boundary)
    lazy = pub.ct_mont_mul(ca, cb)
    assert torch.equal(cs, pub.ct_retag(lazy, -1, 0, out=lazy))
    del lazy, cs
    # ct * k: D(E(a)^k) = k a mod n with 53-bit multipliers
    k = rng.integers(1, 1 << 53, size=N, dtype=np.uint64)
    kw = np.stack([(k & 0xFFFFFFFF).astype(np.uint32), (k >> np.uint64(32)).astype(np.uint32)], axis=1)
    p = priv.decrypt(pub.ct_mul(ca, engine.to_device_words(kw, pub.device), 53))
    torch.cuda.synchronize()
    sh, ph = engine.to_host_words(s), engine.to_host_words(p)
    idx = np.concatenate([np.arange(0, 2048), np.linspace(2048, N - 1, 2048).astype(np.int64)])
    ai, bi = engine.words_to_ints(ra[idx]), engine.words_to_ints(rb[idx])
    assert engine.words_to_ints(sh[idx]) == [(x + y) % key.n for x, y in zip(ai, bi)]
    assert engine.words_to_ints(ph[idx]) == [int(kk) * x % key.n for kk, x in zip(k[idx], ai)]
    # whole-batch checksum: the sum of all decrypted residues is linear too (mod n)
    def total(words):
        acc = 0
        for col in range(words.shape[1] - 1, -1, -1):
            acc = (acc << 32) + int(words[:, col].astype(np.uint64).sum())
        return acc % key.n
    assert total(sh) == (total(ra) + total(rb)) % key.n


@pytest.mark.parametrize("bits,total_n", [(3072, 1 << 20), (4096, 1 << 18)])
def test_config3_4_every_rank_shard_at_full_size(bits, total_n):
    """Configs 4/5 shard the batch over 8 GPUs (block partition, no exchange inside an operation).  The one GPU of the
    test box runs the shards of ALL eight ranks one after the other — the whole configuration at its full size — with,
    per shard: the round trip of every element, ciphertext bits of a sample against the C oracle (explicit
    randomness), and over the whole batch: the product of all ciphertexts (pai_ct_prod per shard, combined on the host —
    what the 8-way final product of SURVEY §8e's reduction does) decrypts to the sum of all residues."""
    key = fixture_key(bits)
    pub, priv = handles(key)
    world = 8
    x_all = np.random.default_rng(1000 + bits).uniform(-1000, 1000, total_n)
    ck = co.COracleKey(key)
    c_enc = ck.ifma_encrypt_djn if co.ifma_available() else ck.encrypt_djn
    total_residues, shard_products = 0, []

    def colsum(words):
        acc = 0
        for col in range(words.shape[1] - 1, -1, -1):
            acc = (acc << 32) + int(words[:, col].astype(np.uint64).sum())
        return acc

    for rank in range(world):
        s0, e0 = sharding.my_shard(total_n, rank, world)
        N = e0 - s0
        x = x_all[s0:e0]
        # raw residues of the SAME exponent class would be needed for a float sum; the check below is on residues mod n
        res, expo = fixedpoint.encode_float64_array(x, key.n, pub.n_words)
        m = engine.to_device_words(res, pub.device)
        r_l = orc.synth_r_limbs(4000 + bits + rank, N, key.randbits)           # explicit randomness: the bits are defined
        ct = pub.encrypt(m, engine.to_device_words(r_l, pub.device))
        back = priv.decrypt(ct)
        torch.cuda.synchronize()
        assert torch.equal(back, m), rank
        assert np.array_equal(fixedpoint.decode_float64_array(engine.to_host_words(back), expo, key.n, key.max_int), x)
        idx = np.linspace(0, N - 1, 24).astype(np.int64)
        got = engine.to_host_words(ct[torch.from_numpy(idx).to(pub.device)])
        assert np.array_equal(got, c_enc(res[idx], r_l[idx])), rank
        # the Python-int oracle itself (not the IFMA port) on 8 elements of EVERY shard — 16 on the first and the last one, the
        # ragged ends of the partition —, CRT decryption on two of them
        for k, j in enumerate(np.linspace(0, len(idx) - 1, 16 if rank in (0, world - 1) else 8).astype(np.int64)):
            i = int(idx[j])
            c = engine.words_to_ints(got[j:j + 1])[0]
            assert c == orc.encrypt(key, engine.words_to_ints(res[i:i + 1])[0], orc.limbs_to_ints(r_l[i:i + 1])[0]), (rank, i)
            if k in (0, 7):
                assert orc.decrypt_crt(key, c) == engine.words_to_ints(res[i:i + 1])[0]
        total_residues += colsum(res)
        shard_products.append(engine.words_to_ints(engine.to_host_words(pub.ct_prod(ct, 1)))[0])
        if rank == 0:
            # homomorphic identities on a shard: D(E(a) E(a)) = 2a and D(E(a)^3) = 3a (mod n)
            s2 = engine.to_host_words(priv.decrypt(pub.ct_add(ct, ct))[:64])
            e3 = torch.full((1, 1), 3, dtype=torch.int32, device=pub.device)
            p3 = engine.to_host_words(priv.decrypt(pub.ct_mul(ct, e3, 2))[:64])
            ai = engine.words_to_ints(res[:64])
            assert engine.words_to_ints(s2) == [2 * a % key.n for a in ai]
            assert engine.words_to_ints(p3) == [3 * a % key.n for a in ai]
        del ct, back, m
    prod = 1
    for c in shard_products:
        prod = prod * c % key.nsq
    assert orc.decrypt_crt(key, prod) == total_residues % key.n


def test_config2_api_level_full_size_negative_multipliers_and_alignment():
    """BASELINE configs[2] at its full size THROUGH THE PUBLIC API with SURVEY §8d's inputs: 2^20 floats, multipliers
    default_rng(2003).uniform(-10, 10) (about half negative: the ciphertext-inversion path of ipcl_python.py:426-437),
    a plaintext addend and a ciphertext addend from default_rng(3003) (exponents differ per element: the alignment of
    ipcl_python.py:570-741), the reference's own composition (E(x) * y + z) (tests/ipcl_python_test.py:40-54).
    Oracle bits on 256 elements spread over the batch, every element decrypted to the reference's 7 decimal places."""
    from pailliercryptolib_python_amd import PaillierPrivateKey, PaillierPublicKey
    from pailliercryptolib_python_amd.bindings import ipclPublicKey

    key = key2048()
    pk = PaillierPublicKey(ipclPublicKey(key.n, 2048, True, hs=key.hs, randbits=key.randbits, device=DEV))
    sk = PaillierPrivateKey(pk, key.p, key.q)
    N = 1 << 20
    x = np.random.default_rng(1002).uniform(-1000, 1000, N)
    y = np.random.default_rng(2003).uniform(-10, 10, N)
    z = np.random.default_rng(3003).uniform(-1000, 1000, N)
    assert 0.45 < (y < 0).mean() < 0.55
    rx, rz = orc.synth_r_limbs(4002, N, key.randbits), orc.synth_r_limbs(4003, N, key.randbits)
    ex = pk.encrypt(x, r=rx)
    ez = pk.encrypt(z, r=rz)
    prod = ex * y                                   # negative multipliers: batch inversion + ct^(n - pt)
    res = prod + z                                  # plaintext addend: raw-encrypt + alignment + product
    both = prod + ez                                # ciphertext addend: alignment + product
    diff = ex - ez                                  # a - b = a + b * (-1.0): every element through the inversion
    idx = np.unique(np.concatenate([np.linspace(0, N - 1, 250).astype(np.int64), np.nonzero(y < 0)[0][:3], np.nonzero(y > 0)[0][:3]]))
    assert len(idx) >= 250

    def sample(enc):
        t = torch.from_numpy(idx).to(enc.words.device)
        return engine.words_to_ints(engine.to_host_words(enc.words[t])), [enc.exponent(int(i)) for i in idx]

    ox = orc.api_encrypt(key, [float(v) for v in x[idx]], orc.limbs_to_ints(rx[idx]))
    oz = orc.api_encrypt(key, [float(v) for v in z[idx]], orc.limbs_to_ints(rz[idx]))
    assert sample(ex) == (ox[0], ox[1])
    op = orc.api_mul_plain(key, *ox, [float(v) for v in y[idx]])
    assert sample(prod) == (op[0], op[1])
    assert sample(res) == tuple(orc.api_add_plain(key, *op, [float(v) for v in z[idx]]))
    assert sample(both) == tuple(orc.api_add_ct(key, *op, *oz))
    assert sample(diff) == tuple(orc.api_sub_ct(key, *ox, *oz))
    # every element, to the reference's assertAlmostEqual precision (7 decimal places)
    for enc, want in ((res, x * y + z), (both, x * y + z), (diff, x - z), (prod, x * y)):
        got = sk.decrypt_to_numpy(enc)
        assert got.shape == (N,) and float(np.abs(got - want).max()) < 5e-8
