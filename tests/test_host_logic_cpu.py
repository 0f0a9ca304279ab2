"""Host-side planning logic that needs no device."""
import re
from pathlib import Path

from pailliercryptolib_python_amd import paillier

ROOT = Path(__file__).resolve().parents[1]


def test_addn_tag_plan():
    """PaillierEncryptedNumber.add_many's tag planning against pai_ct_addn's precondition (csrc/paillier_capi.hip:
    |1 + dom_out - c| <= RPOW_SPAN over the tags c a tile passes through): for every legal combination of the running sum's tag,
    the operands' common tag and the chunk length, either the natural plan fits or the re-plan with the operands retagged to the
    wire form does — ADVICE r05: sixteen operands at tag -3 used to raise PAI_E_INVALID."""
    src = (ROOT / "pailliercryptolib_python_amd" / "csrc" / "kernels_paillier.hpp").read_text()
    span = int(re.search(r"constexpr int RPOW_SPAN = (\d+);", src).group(1))
    assert span == paillier.ADDN_RPOW_SPAN
    capi = (ROOT / "pailliercryptolib_python_amd" / "csrc" / "dispatch_add.hpp").read_text()
    assert "std::min(tag0, 1) + (k - 1) * std::min(tag - 1, 0)" in capi and "std::max(tag0, 1) + (k - 1) * std::max(tag - 1, 0)" in capi
    bad_natural = 0
    for first_chunk in (True, False):
        tag0_range = range(-paillier.DOM_MAX, paillier.DOM_MAX + 1) if first_chunk else \
            range(-paillier.ADDN_ACC_DOM_MAX, paillier.ADDN_ACC_DOM_MAX + 1)
        for tag0 in tag0_range:
            for tag in range(-paillier.DOM_MAX, paillier.DOM_MAX + 1):
                for k in range(2, 17):
                    for last in (False, True):
                        d = paillier._addn_dom_out(tag0, tag, k, last)
                        assert d == 0 if last else abs(d) <= paillier.ADDN_ACC_DOM_MAX
                        if paillier._addn_tags_fit(tag0, tag, k, d):
                            continue
                        bad_natural += 1
                        d = paillier._addn_dom_out(tag0, 0, k, last)
                        assert paillier._addn_tags_fit(tag0, 0, k, d), (tag0, tag, k, last)
    assert bad_natural > 0
    # the case of the finding: sixteen operands at tag -3
    assert not paillier._addn_tags_fit(-3, -3, 16, 0)
    # sixteen fresh ciphertexts keep their natural tag between chunks (no fix-up product)
    assert paillier._addn_dom_out(0, 0, 16, False) == -15 and paillier._addn_tags_fit(0, 0, 16, -15)


def test_host_exponents_equal_the_codec():
    """fixedpoint.float64_exponents / float64_exponents_at (what small float batches use instead of reading the device codec's
    exponents back) against the host codec the goldens pin (encode_float64_array, align_encoded; reference fixedpoint.py:54-96)."""
    import numpy as np

    from pailliercryptolib_python_amd import fixedpoint as fp

    rng = np.random.default_rng(5)
    x = np.concatenate([rng.uniform(-1e6, 1e6, 300), rng.uniform(-1, 1, 100) * 10.0 ** rng.integers(-250, 250, 100),
                        np.array([0.0, -0.0, 1e-200, 9.9e-201, -9.9e-201, 5e-324, 1.0, -1.0, 2.0 ** 52, 2.0 ** 53, 1.7e308])])
    n = (1 << 2047) + 12345
    max_int = n // 3 - 1
    res, expo = fp.encode_float64_array(x, n, 64)
    assert np.array_equal(fp.float64_exponents(x), expo)
    for n_bits_n in (n, (1 << 255) + 7):
        nw = (n_bits_n.bit_length() + 31) // 32
        res, expo = fp.encode_float64_array(x, n_bits_n, nw)
        for tgt in (np.array([int(expo.max())], dtype=np.int32), (expo + rng.integers(-3, 40, expo.shape[0])).astype(np.int32),
                    np.array([2000], dtype=np.int32)):
            _, want = fp.align_encoded(res, expo, tgt, n_bits_n, n_bits_n // 3 - 1)
            assert np.array_equal(fp.float64_exponents_at(x, tgt, n_bits_n.bit_length()), want)


def test_msb_first_product_model_bounds():
    """The cell-exact model of the most-significant-limb-first product (csrc/mont_msb.hpp, tools/msb_model.py): on the four key
    sizes' geometries — random, all-ones-limb and unreduced operands, random and extreme moduli — no lane's own 64-bit cell wraps,
    every quotient digit is the true one or one below, the accumulator stays below 2 Mt, and the product is a b mod M.  (The kernel
    itself is held to CPython on the GPU: tests/test_gpu_paillier_abi.py::test_ct_add_by_one_msb_first_product.)"""
    import importlib.util
    import random
    from pathlib import Path

    spec = importlib.util.spec_from_file_location("msb_model", Path(__file__).resolve().parent.parent / "tools" / "msb_model.py")
    mm = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mm)
    rng = random.Random(5)
    for key, (NLL, T, U) in mm.GEOS.items():
        for M in (mm.rand_modulus(2 * key, rng), (1 << (2 * key)) - 1 - 2 * rng.getrandbits(40), (1 << (2 * key - 1)) + 1 + 2 * rng.getrandbits(40)):
            p = mm.Params(M, NLL, T, U)
            assert p.ok
            full = (1 << (2 * key)) - 1
            ones = (1 << (M.bit_length() - 1)) - 1
            stats = {}
            for a, b in ((M - 1, M - 1), (1, 1), (0, 5), (full, full), (ones, ones), (rng.randrange(M), rng.randrange(M)), (full, rng.getrandbits(2 * key))):
                assert mm.msb_mul(p, a, b, stats) == a * b % M
            assert set(stats) <= {0, 1, "qmax"}
    # the corners of the context's conditions (csrc/paillier_capi.hip: build_msb_ctx): 3 and 26 bits of the modulus in its top limb,
    # shifted up by 1 and by 16 limbs, smallest and largest modulus of the bit length
    for NLL, T, U in ((36, 4, 6), (28, 8, 4)):
        NL = NLL * T
        for off in (1, 16):
            for tb in (3, 26):
                bits = 29 * (NL - 1 - off) + tb
                for M in ((1 << bits) - 1 - 2 * rng.getrandbits(30), (1 << (bits - 1)) + 1 + 2 * rng.getrandbits(30)):
                    p = mm.Params(M, NLL, T, U)
                    assert p.ok and (p.tb, p.off) == (tb, off)
                    full = (1 << bits) - 1
                    for a, b in ((M - 1, M - 1), (full, full), (rng.randrange(M), rng.randrange(M))):
                        assert mm.msb_mul(p, a, b) == a * b % M
