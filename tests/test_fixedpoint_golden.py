"""CPU: the oracle's codec restatement AND the product's host codec (scalar + vectorised) against golden
vectors produced by the reference's own fixedpoint.py (tests/golden/make_fixedpoint_golden.py)."""
import json
from pathlib import Path

import numpy as np
import pytest

from oracle import paillier_oracle as orc
from pailliercryptolib_python_amd import fixedpoint as fp

G = json.loads((Path(__file__).parent / "golden" / "fixedpoint_golden.json").read_text())
N_KEY, MAX_INT = int(G["n"], 16), int(G["max_int"], 16)
CAST = {"float": float, "int": int, "bool": lambda s: bool(int(s)), "np.float64": np.float64, "np.float32": np.float32,
        "np.int64": np.int64, "np.int32": np.int32, "np.int16": np.int16, "np.int8": np.int8, "np.uint8": np.uint8}
ERR = {"ValueError": ValueError, "OverflowError": OverflowError, "TypeError": TypeError}


def value_of(rec):
    t, v = rec["t"], rec["v"]
    if "float" in t:
        x = float("nan") if v == "nan" else float.fromhex(v) if v not in ("inf", "-inf") else float(v)
        return CAST[t](x)
    return CAST[t](int(v)) if t != "bool" else bool(int(v))


def dec_of(s):
    return int(s[4:]) if s.startswith("int:") else float.fromhex(s)


def same(a, b):
    return type(a) is type(b) and (a == b or (a != a and b != b))


def test_golden_matches_bench_key():
    assert N_KEY == orc.BENCH_P * orc.BENCH_Q and MAX_INT == N_KEY // 3 - 1


@pytest.mark.parametrize("impl", ["oracle", "product"])
def test_encode_scalar_against_reference_vectors(impl):
    for rec in G["encode"]:
        v = value_of(rec)
        if "err" in rec:
            with pytest.raises(ERR[rec["err"]]):
                with np.errstate(all="ignore"):
                    (orc.fp_encode(v, N_KEY, MAX_INT) if impl == "oracle" else fp.FixedPointNumber.encode(v, N_KEY, MAX_INT))
            continue
        if impl == "oracle":
            enc, ex = orc.fp_encode(v, N_KEY, MAX_INT)
            dec = orc.fp_decode(enc, ex, N_KEY, MAX_INT)
        else:
            e = fp.FixedPointNumber.encode(v, N_KEY, MAX_INT)
            enc, ex, dec = e.encoding, e.exponent, e.decode()
        assert enc == int(rec["enc"], 16) and ex == rec["exp"], rec
        assert same(dec, dec_of(rec["dec"])), rec


@pytest.mark.parametrize("impl", ["oracle", "product"])
def test_decode_against_reference_vectors(impl):
    for rec in G["decode"]:
        enc, ex = int(rec["enc"], 16), rec["exp"]
        call = (lambda: orc.fp_decode(enc, ex, N_KEY, MAX_INT)) if impl == "oracle" else \
               (lambda: fp.FixedPointNumber(enc, ex, N_KEY, MAX_INT).decode())
        if "dec_err" in rec:
            with pytest.raises(ERR[rec["dec_err"]]):
                call()
        else:
            assert same(call(), dec_of(rec["dec"])), rec


def test_vectorised_encode_matches_reference_vectors():
    recs = [r for r in G["encode"] if r["t"] == "float" and "err" not in r]
    x = np.array([value_of(r) for r in recs], dtype=np.float64)
    res, expo = fp.encode_float64_array(x, N_KEY, 64)
    got = [int.from_bytes(row.tobytes(), "little") for row in res]
    assert got == [int(r["enc"], 16) for r in recs]
    assert expo.tolist() == [r["exp"] for r in recs]
    # generic entry point: float list, int list, mixed list
    res2, expo2 = fp.encode_array([float(v) for v in x], N_KEY, MAX_INT, 64)
    assert np.array_equal(res, res2) and np.array_equal(expo, expo2)
    ints = [r for r in G["encode"] if r["t"] == "int" and "err" not in r]
    res3, expo3 = fp.encode_array([int(r["v"]) for r in ints], N_KEY, MAX_INT, 64)
    assert [int.from_bytes(row.tobytes(), "little") for row in res3] == [int(r["enc"], 16) for r in ints]
    assert expo3.tolist() == [0] * len(ints)


def test_vectorised_encode_large_random_equals_scalar_definition():
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.uniform(-1000, 1000, 3000), rng.normal(0, 1, 1000) * 10.0 ** rng.integers(-40, 40, 1000)])
    res, expo = fp.encode_float64_array(x, N_KEY, 64)
    for i in range(0, x.shape[0], 7):
        enc, ex = orc.fp_encode(float(x[i]), N_KEY, MAX_INT)
        assert int.from_bytes(res[i].tobytes(), "little") == enc and expo[i] == ex


def test_vectorised_decode_paths():
    rng = np.random.default_rng(6)
    x = rng.uniform(-1000, 1000, 500)
    res, expo = fp.encode_float64_array(x, N_KEY, 64)
    vals = fp.decode_array(res, expo, N_KEY, MAX_INT)
    assert all(type(v) is float for v in vals) and np.array_equal(np.array(vals), x)
    assert np.array_equal(fp.decode_float64_array(res, expo, N_KEY, MAX_INT), x)
    # integers decode to Python ints (exponent 0), and mantissas beyond 2^63 take the exact path
    big = [10**30, -(10**30), 5, -7]
    r2, e2 = fp.encode_array(big, N_KEY, MAX_INT, 64)
    assert fp.decode_array(r2, e2, N_KEY, MAX_INT) == big
    assert np.array_equal(fp.decode_float64_array(r2, e2, N_KEY, MAX_INT), np.array([1e30, -1e30, 5.0, -7.0]))
    with pytest.raises(OverflowError):
        fp.decode_array(np.frombuffer((N_KEY // 2).to_bytes(256, "little"), dtype="<u4")[None, :], [0], N_KEY, MAX_INT)


def test_nonfinite_inputs_raise_like_the_reference():
    with pytest.raises(ValueError):
        fp.encode_float64_array(np.array([1.0, float("nan")]), N_KEY, 64)
    with pytest.raises(OverflowError):
        fp.encode_float64_array(np.array([float("inf")]), N_KEY, 64)


def test_align_encoded_equals_raising_the_raw_encryption():
    """fixedpoint.align_encoded (host counterpart of pai_fp_encode_at): encoding a plaintext AT a larger target exponent
    equals raising its raw encryption to 2^(target - exponent) — (1 + m n)^(2^d) = 1 + (m 2^d mod n) n — for positive,
    negative, zero, integer and huge-shift cases; elements whose shifted magnitude would not stay below n keep their
    exponent (the ciphertext path raises those)."""
    import numpy as np

    from oracle import paillier_oracle as orc
    from pailliercryptolib_python_amd import fixedpoint as fp

    key = orc.make_key(orc.BENCH_P, orc.BENCH_Q, djn_x=3, bits=2048)
    n, nsq = key.n, key.nsq
    vals = [1.5, -2.25, 0.0, 3, -7, 1e10, -1e-5, 2.0 ** 60, -(2.0 ** -40), 123456789, 5e-324]
    res, ex = fp.encode_array(vals, n, key.max_int, 64)
    for tg in (np.array([60] * len(vals)), np.array([60, 60, 5, 10, 3, 10, 100, -8, 200, 1, 7]), np.array([1990]), np.array([5000])):
        r2, e2 = fp.align_encoded(res, ex, tg, n, key.max_int)
        t = np.broadcast_to(tg, ex.shape)
        for i in range(len(vals)):
            enc = int.from_bytes(res[i].tobytes(), "little")
            enc2 = int.from_bytes(r2[i].tobytes(), "little")
            d = int(e2[i]) - int(ex[i])
            assert d >= 0 and int(e2[i]) in (int(ex[i]), int(t[i]))
            assert pow(orc.raw_encrypt(enc, n), 1 << d, nsq) == orc.raw_encrypt(enc2, n), (i, int(t[i]))
            if enc == 0 and t[i] > ex[i]:
                assert e2[i] == t[i]                                  # zero takes any larger target
            if 0 < d < 900:                                           # a moved element still decodes to the same value
                assert orc.fp_decode(enc2, int(e2[i]), n, key.max_int) == orc.fp_decode(enc, int(ex[i]), n, key.max_int)
    # a shift that would pass n is refused (exponent kept)
    r3, e3 = fp.align_encoded(res[:1], ex[:1], np.array([int(ex[0]) + 2040]), n, key.max_int)
    assert e3[0] == ex[0] and np.array_equal(r3[0], res[0])
