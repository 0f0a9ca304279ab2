"""API transcripts: small programs over the public API, written once and run by three backends.

  * tests/golden/make_api_transcripts.py runs them through the REFERENCE's own Python layer
    (/root/reference/src/ipcl_python/ipcl_python.py, imported by path in the build container only) and records, for
    every named result, (exponents, ciphertext integers, decrypted values) in tests/golden/api_transcripts.json;
  * tests/test_api_transcripts.py (CPU) replays them on the oracle's restatement (oracle/paillier_oracle.py: api_*);
  * tests/test_gpu_transcripts.py (GPU) replays them on the product's public API and compares bit for bit.

A program takes a backend ``B`` with ``B.enc(values, seed)`` (encrypt with the obfuscator randomness
``synth_r(seed, len, randbits)`` injected), ``B.raw(values)`` (raw_encrypt) and ``B.obf(en, seed)`` (re-obfuscate in
place); everything else is Python operators and methods on whatever encrypted-number type the backend returns, i.e. the
reference's operator surface (ipcl_python.py:365-410,746-775,882-930).  Inputs are literals or seeded numpy draws, so the
JSON needs to hold outputs only.  This file is test code of this repository; nothing of the reference is in it.
"""
from __future__ import annotations

import numpy as np

DBL_MAX = 1.7976931348623157e308


def synth_r(seed: int, count: int, randbits: int):
    """Obfuscator randomness of one encrypt call: `count` integers below 2^randbits from a seeded numpy generator."""
    words = (randbits + 31) // 32
    a = np.random.default_rng(seed).integers(0, 1 << 32, size=(count, words), dtype=np.uint32)
    top = randbits - 32 * (words - 1)
    if top < 32:
        a[:, -1] &= np.uint32((1 << top) - 1)
    return [int.from_bytes(row.astype("<u4").tobytes(), "little") for row in a]


# ---- programs ---------------------------------------------------------------------------------------------------------
def p_encrypt_edges(B):
    """SURVEY App. C edge list (everything the codec accepts): signed zero, subnormals -> 0, the 1e-200 threshold, 2^53,
    2^60, -2^70, DBL_MAX, bool, numpy scalars, 10^30."""
    vals = [0.0, -0.0, 5e-324, 2.2e-308, 9.99e-201, 1e-200, 2.0 ** 53, 2.0 ** 60, -(2.0 ** 70), DBL_MAX, True, np.int16(-3),
            np.int32(70000), np.int64(-(1 << 40)), np.float32(0.1), np.float64(-2.5), 10 ** 30, -(10 ** 30), 1234.5678, -5111.2834]
    return {"enc": B.enc(vals, 101), "raw": B.raw(vals), "one": B.enc(3.25, 102), "one_int": B.enc(-17, 103)}


def p_mul_chain(B):
    """(E(x) * y + z) * t of the reference's tests/ipcl_python_test.py:40-54 with negative y (ciphertext inversion)."""
    n = 8
    x = np.ones(n) * 37
    y = np.ones(n) * 613 * -1
    z = np.ones(n) * 0.37124
    t = list(range(n))
    en_x = B.enc(x, 201)
    step1 = en_x * y
    step2 = step1 + z
    return {"x": en_x, "xy": step1, "xy_z": step2, "res": step2 * t}


def p_add_chain(B):
    """en_x + en_y + en_z + en_t of tests/ipcl_python_test.py:24-38 (float arrays, a float fraction, an int list)."""
    n = 6
    en = [B.enc(np.ones(n) * 41, 301), B.enc(np.ones(n) * 977, 302), B.enc(np.ones(n) * 0.6180339887, 303), B.enc(list(range(n)), 304)]
    return {"res": en[0] + en[1] + en[2] + en[3]}


def p_scalar_loop(B):
    """en_x = en_x + 5000; en_x = en_x - 0.2, three times (tests/ipcl_python_test.py:56-66)."""
    en = B.enc(9, 401)
    out = {}
    for i in range(3):
        en = en + 5000
        out[f"plus{i}"] = en
        en = en - 0.2
        out[f"minus{i}"] = en
    return out


def p_sub_div(B):
    rng = np.random.default_rng(501)
    n = 5
    a_v, b_v = rng.uniform(-1000, 1000, n), rng.uniform(-1e-3, 1e-3, n)
    a, b = B.enc(a_v, 502), B.enc(b_v, 503)
    pl = rng.uniform(-50, 50, n)
    return {
        "a_minus_b": a - b, "b_minus_a": b - a, "a_minus_arr": a - pl, "a_minus_list": a - [float(v) for v in pl],
        "list_minus_a": [float(v) for v in pl] - a, "scalar_minus_a": 7.5 - a, "int_plus_a": 5 + a, "a_plus_arr": a + pl,
        "a_times_neg": a * -2.5, "a_times_int": a * 3, "neg_times_a": -0.125 * a, "a_times_zero": a * 0,
        "a_div_4": a / 4.0, "a_div_arr": a / np.array([2.0, -4.0, 0.5, 8.0, -3.0]), "a_div_list": a / [3.0, 7.0, -1.5, 2.0, 10.0],
        "a_times_arr": a * np.array([1.5, -2.25, 0.0, 1e6, -1e-6]), "a_times_ints": a * [1, -2, 3, -4, 5],
    }


def p_broadcast(B):
    """A length-1 ciphertext on either side of + (ipcl_python.py:365-375,588-660: the broadcast branch of alignment)."""
    a = B.enc([1.5, -2.0, 1000.25, 3, 1e-4], 601)
    one = B.enc(0.75, 602)
    big = B.enc(1 << 40, 603)
    return {"a_plus_one": a + one, "one_plus_a": one + a, "a_plus_big": a + big, "a_minus_one": a - one, "one_plus_one": one + big}


def p_matmul(B):
    rng = np.random.default_rng(701)
    out = {}
    for tag, (m, n, k) in {"232": (2, 3, 2), "312": (3, 1, 2), "143": (1, 4, 3)}.items():
        x, y = rng.uniform(-4, 4, (m, n)), rng.uniform(-4, 4, (n, k))
        out[f"mm_{tag}"] = B.enc(x.flatten(), 710 + m * 100 + n * 10 + k) @ y
        out[f"rmm_{tag}"] = x.tolist() @ B.enc(y.flatten(), 720 + m * 100 + n * 10 + k)
    v = rng.uniform(-4, 4, 3)
    out["mm_1d"] = B.enc(rng.uniform(-4, 4, 6), 731) @ v                 # (2 x 3) @ (3,)
    out["rmm_1d"] = [float(t) for t in v] @ B.enc(rng.uniform(-4, 4, 6), 732)   # (3,) @ (3 x 2)
    en = B.enc(rng.uniform(-4, 4, 4), 733)
    en @= rng.uniform(-4, 4, (2, 2))
    out["imm_22"] = en
    out["mm_ints"] = B.enc([1, -2, 3, 4], 734) @ np.array([[2.0, -1.0], [0.5, 3.0]])
    return out


def p_reductions_len1(B):
    """sum / mean / dot: the reference's sum() only works for one element (it hands a Python list to __padded_ct,
    SURVEY App. B), so the transcripts hold length 1; longer reductions are pinned to the oracle's restatement only."""
    a = B.enc(-12.625, 801)
    return {"sum": a.sum(), "mean": a.mean(), "dot": a.dot([3.5]), "dot_neg": a.dot(np.array([-0.25]))}


def p_container(B):
    a = B.enc([1.0, -2.0, 3.5, 4, -5e-3, 6e3], 901)
    b = B.enc([10.0, 20.0], 902)
    parts = {"slice": a[1:4], "item": a[2], "last": a[5], "slice_plus": a[2:4] + b, "iter_sum3": None}
    it = iter(a)
    acc = next(it)
    acc = acc + next(it)
    acc = acc + next(it)
    parts["iter_sum3"] = acc
    return parts


def p_obfuscate(B):
    a = B.raw([1.25, -7, 0.0])
    B.obf(a, 1001)
    b = B.enc([2.5, 3], 1002)
    B.obf(b, 1003)
    return {"raw_then_obf": a, "enc_then_obf": b, "sum": a[0:2] + b}


PROGRAMS = {
    "encrypt_edges": p_encrypt_edges, "mul_chain": p_mul_chain, "add_chain": p_add_chain, "scalar_loop": p_scalar_loop,
    "sub_div": p_sub_div, "broadcast": p_broadcast, "matmul": p_matmul, "reductions_len1": p_reductions_len1,
    "container": p_container, "obfuscate": p_obfuscate,
}
KEY_BITS = (2048, 1024)        # fixture keys of tests/golden/fixture_keys.json (2048 = the reference's bench constants)
DJN_X = 0x1234567


# ---- serialisation of results -----------------------------------------------------------------------------------------
def ser_value(v):
    if isinstance(v, (int, np.integer)) and not isinstance(v, bool):
        return "int:%d" % int(v)
    return "float:" + float(v).hex()


def record(expo, cts, dec):
    return {"expo": [int(e) for e in expo], "ct": ["%x" % int(c) for c in cts], "dec": [ser_value(v) for v in dec]}


def run_program(name, B):
    """{result name: record} of one program on backend B (B.dump(en) -> (exponents, ciphertext ints, decrypted list))."""
    return {k: record(*B.dump(en)) for k, en in PROGRAMS[name](B).items()}


# ---- the oracle as a backend (used by the CPU test; an encrypted number is a (ciphertexts, exponents) pair) -------------
class OracleEN:
    """(ciphertext ints, exponents) with the operator surface of PaillierEncryptedNumber, every operator delegating to
    the oracle's restatement of the composition (oracle/paillier_oracle.py: api_*)."""

    def __init__(self, key, cts, expo):
        self.key, self.c, self.e = key, list(cts), list(expo)

    def _w(self, pair):
        return OracleEN(self.key, *pair)

    def __len__(self):
        return len(self.c)

    def __add__(self, other):
        from oracle import paillier_oracle as orc
        if isinstance(other, OracleEN):
            if len(self) == 1 and len(other) > 1:
                return other + self
            return self._w(orc.api_add_ct(self.key, self.c, self.e, other.c, other.e))
        return self._w(orc.api_add_plain(self.key, self.c, self.e, other))

    __radd__ = __add__

    def __mul__(self, other):
        from oracle import paillier_oracle as orc
        return self._w(orc.api_mul_plain(self.key, self.c, self.e, other if np.isscalar(other) else list(other)))

    __rmul__ = __mul__

    def __sub__(self, other):
        if isinstance(other, list):
            other = np.array(other)
        return self + (other * -1.0)

    def __rsub__(self, other):
        return (self * -1.0) + other

    def __truediv__(self, other):
        from oracle import paillier_oracle as orc
        return self._w(orc.api_truediv(self.key, self.c, self.e, other))

    def __matmul__(self, other):
        from oracle import paillier_oracle as orc
        return self._w(orc.api_matmul(self.key, self.c, self.e, other))

    def __rmatmul__(self, other):
        from oracle import paillier_oracle as orc
        return self._w(orc.api_matmul(self.key, self.c, self.e, other, rhs=True))

    def sum(self):
        from oracle import paillier_oracle as orc
        return self._w(orc.api_sum(self.key, self.c, self.e))

    def mean(self):
        from oracle import paillier_oracle as orc
        return self._w(orc.api_mean(self.key, self.c, self.e))

    def dot(self, other):
        from oracle import paillier_oracle as orc
        return self._w(orc.api_dot(self.key, self.c, self.e, list(other)))

    def __getitem__(self, k):
        if isinstance(k, int):
            k = slice(k, k + 1)
        return OracleEN(self.key, self.c[k], self.e[k])

    def __iter__(self):
        return (self[i] for i in range(len(self)))


class OracleBackend:
    def __init__(self, key):
        self.key = key

    def _vals(self, values):
        return [values] if np.isscalar(values) else list(values)

    def enc(self, values, seed):
        from oracle import paillier_oracle as orc
        v = self._vals(values)
        return OracleEN(self.key, *orc.api_encrypt(self.key, v, synth_r(seed, len(v), self.key.randbits)))

    def raw(self, values):
        from oracle import paillier_oracle as orc
        return OracleEN(self.key, *orc.api_encrypt(self.key, self._vals(values), None))

    def obf(self, en, seed):
        from oracle import paillier_oracle as orc
        en.c = [orc.apply_obfuscator(self.key, c, r) for c, r in zip(en.c, synth_r(seed, len(en), self.key.randbits))]

    def dump(self, en):
        from oracle import paillier_oracle as orc
        return en.e, en.c, orc.api_decrypt(self.key, en.c, en.e)
