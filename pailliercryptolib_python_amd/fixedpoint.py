"""Fixed-point codec: float/int <-> (mantissa mod n, base-2 exponent).

Host-side mirror of the part of ``src/ipcl_python/bindings/fixedpoint.py`` that the reference API
actually calls: ``FixedPointNumber.encode`` (``:54-96``), ``.decode`` (``:98-115``), the constructor
(``:35-47``) and the constants ``BASE``/``FLOAT_MANTISSA_BITS`` (``:29-31``).  The operator overloads
and ``FixedPointEndec`` of that file are never used by ``ipcl_python.py`` and are out of scope
(SURVEY.md §2 row 2).

The reference encodes one Python object at a time (6.5 us per element measured); the array paths
here (`encode_array`, `decode_array`) are vectorised with numpy and produce/consume the flat
``[N][words]`` little-endian limb matrices of the C ABI.  They are bit-identical to the scalar
definition (checked against golden vectors produced by the reference's own file:
``tests/golden/fixedpoint_golden.json``).
"""
from __future__ import annotations

import math
import sys
from typing import List, Sequence, Tuple

import numpy as np


class FixedPointNumber(object):
    """Scalar codec object with the reference's interface (encoding, exponent, n, max_int)."""

    BASE = 2
    LOG2_BASE = math.log(BASE, 2)
    FLOAT_MANTISSA_BITS = sys.float_info.mant_dig

    def __init__(self, encoding, exponent, n, max_int=None):
        self.n = n
        self.max_int = n // 2 if max_int is None else max_int
        self.encoding = encoding
        self.exponent = exponent

    @classmethod
    def encode(cls, scalar, n, max_int=None):
        """(scalar) -> FixedPointNumber.  Same rules, same exceptions as fixedpoint.py:54-96 for
        precision=None / max_exponent=None (the only way ipcl_python.py calls it)."""
        if max_int is None:
            max_int = n // 2
        if np.abs(scalar) < 1e-200:                       # :64-65
            scalar = 0
        if isinstance(scalar, (int, np.int16, np.int32, np.int64)):      # :72-74
            exponent = 0
        elif isinstance(scalar, (float, np.float16, np.float32, np.float64)):   # :75-79
            scalar = float(scalar)     # numpy>=2 would overflow float16/32 * 2**k; numpy 1.23 (pinned upstream) promoted
            exponent = cls.FLOAT_MANTISSA_BITS - math.frexp(scalar)[1]
        else:
            raise TypeError("Don't know the precision of type %s." % type(scalar))
        int_fixpoint = int(round(scalar * pow(cls.BASE, exponent)))     # :89
        if abs(int_fixpoint) > max_int:                   # :91-94
            raise ValueError(
                f"Integer needs to be within +/- {max_int},but got {int_fixpoint},"
                f"basic info, scalar={scalar}, base={cls.BASE}, exponent={exponent}"
            )
        return cls(int_fixpoint % n, exponent, n, max_int)

    def decode(self):
        """fixedpoint.py:98-115: int when exponent <= 0, float otherwise."""
        if self.encoding >= self.n:
            raise ValueError("Attempted to decode corrupted number")
        elif self.encoding <= self.max_int:
            mantissa = self.encoding
        elif self.encoding >= self.n - self.max_int:
            mantissa = self.encoding - self.n
        else:
            raise OverflowError(
                f"Overflow detected in decode number, encoding: {self.encoding}, {self.exponent} {self.n}"
            )
        return mantissa * pow(self.BASE, -self.exponent)


# ---------------------------------------------------------------------------------------------------
# array paths
# ---------------------------------------------------------------------------------------------------
_MANT = FixedPointNumber.FLOAT_MANTISSA_BITS


def _words_of(v: int, words: int) -> np.ndarray:
    return np.frombuffer(int(v).to_bytes(4 * words, "little"), dtype="<u4").copy()


def is_float_batch(values) -> bool:
    """True for what the vectorised / device float paths accept: 1-D float ndarrays and non-empty lists of
    Python floats."""
    if isinstance(values, np.ndarray):
        return values.dtype in (np.float64, np.float32, np.float16) and values.ndim == 1 and values.shape[0] > 0
    return isinstance(values, (list, tuple)) and len(values) > 0 and all(type(v) is float for v in values)


def checked_float64(values) -> np.ndarray:
    """Contiguous float64 copy of a float batch; NaN / infinity raise what the reference's int(round(..)) raises
    (fixedpoint.py:89)."""
    x = np.ascontiguousarray(np.asarray(values, dtype=np.float64))
    if not np.all(np.isfinite(x)):
        bad = x[~np.isfinite(x)][0]
        if np.isnan(bad):
            raise ValueError("cannot convert float NaN to integer")
        raise OverflowError("cannot convert float infinity to integer")
    return x


def decode_mantissas(mant: np.ndarray, exponents) -> List:
    """int64 mantissas + exponents -> list with the reference's element types (fixedpoint.py:115
    ``mantissa * pow(2, -exponent)``): Python int when exponent <= 0, float otherwise."""
    e = np.asarray(exponents, dtype=np.int64)
    fl = mant.astype(np.float64) * np.ldexp(1.0, (-e).clip(-2000, 2000).astype(np.int32))     # == float(m) * 2.0**-e
    out = fl.tolist()
    for i in np.nonzero(e <= 0)[0].tolist():
        out[i] = int(mant[i]) * (1 << int(-e[i]))
    return out


def float64_mantissas(x: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """(signed 53-bit mantissas int64[N], exponents int32[N]) of finite doubles: the integer the reference's
    ``int(round(scalar * 2**exponent))`` yields (fixedpoint.py:75-89), before the reduction modulo n."""
    tiny = np.abs(x) < 1e-200
    man, ex = np.frexp(x)
    expo = (_MANT - ex).astype(np.int32)
    mant = np.ldexp(man, _MANT).astype(np.int64)          # exact: |mant| < 2^53
    expo[tiny] = 0
    mant[tiny] = 0
    return mant, expo


def float64_exponents(x: np.ndarray) -> np.ndarray:
    """The exponents float64_mantissas / the device codec (csrc/kernels_codec.hpp: k_fp_encode_f64) assign to finite doubles."""
    expo = (_MANT - np.frexp(x)[1]).astype(np.int32)
    expo[np.abs(x) < 1e-200] = 0
    return expo


def float64_exponents_at(x: np.ndarray, target: np.ndarray, n_bits: int) -> np.ndarray:
    """The exponents pai_fp_encode_at (kernels_codec.hpp: k_fp_encode_at) assigns: an element whose own exponent is below its
    target moves to the target when it is zero or when its 53-bit mantissa shifted by the distance stays below 2^(bits(n) - 2)."""
    e0 = float64_exponents(x).astype(np.int64)
    t = np.broadcast_to(np.asarray(target, dtype=np.int64).reshape(-1), e0.shape)
    dist = t - e0
    move = (dist > 0) & ((np.abs(x) < 1e-200) | (_MANT + dist <= n_bits - 2))
    return np.where(move, t, e0).astype(np.int32)


def encode_float64_array(x: np.ndarray, n: int, n_words: int) -> Tuple[np.ndarray, np.ndarray]:
    """float64[N] -> (residues uint32[N][n_words], exponents int32[N]); requires n > 2^66."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    if not np.all(np.isfinite(x)):
        bad = x[~np.isfinite(x)][0]
        if np.isnan(bad):
            raise ValueError("cannot convert float NaN to integer")
        raise OverflowError("cannot convert float infinity to integer")
    if n.bit_length() <= 66:
        raise ValueError("vectorised encode needs a modulus of more than 66 bits")
    tiny = np.abs(x) < 1e-200
    man, ex = np.frexp(x)
    expo = (_MANT - ex).astype(np.int32)
    mant = np.ldexp(man, _MANT).astype(np.int64)          # exact: |mant| < 2^53
    expo[tiny] = 0
    mant[tiny] = 0
    neg = mant < 0
    a = np.abs(mant).astype(np.uint64)
    N = x.shape[0]
    out = np.zeros((N, n_words), dtype=np.uint32)
    # non-negative: the mantissa itself
    out[:, 0] = (a & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    out[:, 1] = (a >> np.uint64(32)).astype(np.uint32)
    if neg.any():
        # n - a with a < 2^53: low 64 bits wrap, a single borrow may reach the upper words
        n_lo = np.uint64(n & 0xFFFFFFFFFFFFFFFF)
        an = a[neg]
        lo = n_lo - an                                     # wraps mod 2^64
        borrow = an > n_lo
        hi_words = _words_of(n >> 64, n_words - 2)
        hi_words_b = _words_of((n >> 64) - 1, n_words - 2)
        rows = np.where(borrow[:, None], hi_words_b[None, :], hi_words[None, :])
        blk = np.empty((an.shape[0], n_words), dtype=np.uint32)
        blk[:, 0] = (lo & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        blk[:, 1] = (lo >> np.uint64(32)).astype(np.uint32)
        blk[:, 2:] = rows
        out[neg] = blk
    return out, expo


def encode_array(values, n: int, max_int: int, n_words: int) -> Tuple[np.ndarray, np.ndarray]:
    """Sequence of ints/floats (list or 1-D ndarray) -> (residues, exponents).

    float64 / float32 / float16 ndarrays and lists of Python floats take the vectorised path; anything
    else (Python ints of arbitrary size, mixed lists, numpy ints) goes element by element through
    the scalar definition so that type checks and errors are the reference's."""
    if isinstance(values, np.ndarray) and values.dtype in (np.float64, np.float32, np.float16) and values.ndim == 1 \
            and n.bit_length() > 66:
        return encode_float64_array(values.astype(np.float64), n, n_words)
    if isinstance(values, (list, tuple)) and len(values) > 0 and all(type(v) is float for v in values) \
            and n.bit_length() > 66:
        return encode_float64_array(np.asarray(values, dtype=np.float64), n, n_words)
    N = len(values)
    out = np.zeros((N, n_words), dtype=np.uint32)
    expo = np.zeros(N, dtype=np.int32)
    step = 4 * n_words
    buf = bytearray(step * N)
    for i, v in enumerate(values):
        enc = FixedPointNumber.encode(v, n, max_int)
        buf[i * step:(i + 1) * step] = enc.encoding.to_bytes(step, "little")
        expo[i] = enc.exponent
    out = np.frombuffer(bytes(buf), dtype="<u4").reshape(N, n_words).copy()
    return out, expo


def align_encoded(residues: np.ndarray, exponents: np.ndarray, target: np.ndarray, n: int, max_int: int):
    """Host counterpart of pai_fp_encode_at for the element-by-element codec path: an encoding whose exponent e is below
    its target t becomes (mantissa * 2^(t - e)) mod n at exponent t whenever |mantissa| 2^(t-e) < 2^(bits(n) - 2) —
    what raising its RAW encryption to 2^(t - e) yields (ipcl_python.py:570-741), computed on the plaintext."""
    residues = np.ascontiguousarray(residues, dtype="<u4").copy()
    expo = np.asarray(exponents, dtype=np.int32).copy()
    tgt = np.broadcast_to(np.asarray(target, dtype=np.int64).reshape(-1), expo.shape)
    step = 4 * residues.shape[1]
    nbits = n.bit_length()
    for i in np.nonzero(tgt > expo)[0].tolist():
        enc = int.from_bytes(residues[i].tobytes(), "little")
        if enc == 0:
            expo[i] = int(tgt[i])
            continue
        mant = enc if enc <= max_int else enc - n
        d = int(tgt[i]) - int(expo[i])
        if enc > max_int and enc < n - max_int:
            continue                                          # overflow zone: leave it to the ciphertext path
        if abs(mant).bit_length() + d <= nbits - 2:
            residues[i] = np.frombuffer(((mant << d) % n).to_bytes(step, "little"), dtype="<u4")
            expo[i] = int(tgt[i])
    return residues, expo


def decode_array(residues: np.ndarray, exponents: Sequence[int], n: int, max_int: int) -> List:
    """(residues uint32[N][n_words], exponents) -> list of decoded values, element types as the
    reference returns them (int when exponent <= 0, float otherwise; fixedpoint.py:115)."""
    residues = np.ascontiguousarray(residues, dtype="<u4")
    raw = residues.tobytes()
    step = 4 * residues.shape[1]
    out = []
    thr = n - max_int
    for i in range(residues.shape[0]):
        enc = int.from_bytes(raw[i * step:(i + 1) * step], "little")
        if enc >= n:
            raise ValueError("Attempted to decode corrupted number")
        elif enc <= max_int:
            mant = enc
        elif enc >= thr:
            mant = enc - n
        else:
            raise OverflowError(f"Overflow detected in decode number, encoding: {enc}, {exponents[i]} {n}")
        out.append(mant * pow(2, -int(exponents[i])))
    return out


def decode_float64_array(residues: np.ndarray, exponents: np.ndarray, n: int, max_int: int) -> np.ndarray:
    """Fast path to a float64 ndarray for mantissas of magnitude < 2^63 (always true right after
    encrypt->decrypt of floats); falls back to the exact element-wise path otherwise."""
    residues = np.ascontiguousarray(residues, dtype=np.uint32)
    N, W = residues.shape
    expo = np.asarray(exponents, dtype=np.int64)
    lo = residues[:, 0].astype(np.uint64) | (residues[:, 1].astype(np.uint64) << np.uint64(32))
    hi_zero = ~residues[:, 2:].any(axis=1)
    pos = hi_zero & (lo < np.uint64(1 << 63))
    n_lo = np.uint64(n & 0xFFFFFFFFFFFFFFFF)
    hi_words = _words_of(n >> 64, W - 2)
    hi_words_b = _words_of((n >> 64) - 1, W - 2)
    eq_hi = (residues[:, 2:] == hi_words[None, :]).all(axis=1)
    eq_hib = (residues[:, 2:] == hi_words_b[None, :]).all(axis=1)
    a = n_lo - lo                                          # |mantissa| mod 2^64 for negatives
    neg = (~pos) & ((eq_hi & (lo <= n_lo)) | (eq_hib & (lo > n_lo))) & (a < np.uint64(1 << 63)) & (a > 0)
    if not np.all(pos | neg):
        vals = decode_array(residues, expo, n, max_int)
        return np.asarray([float(v) for v in vals], dtype=np.float64)
    mant = np.where(pos, lo.astype(np.float64), -(a.astype(np.float64)))
    return np.ldexp(mant, (-expo).astype(np.int64).astype(np.int32))
