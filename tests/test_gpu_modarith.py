"""GPU parity of the generic modular kernels (pai_modmul / pai_modexp_fixed / pai_modexp_var) against
CPython big-integer arithmetic, through the C ABI.  Bit-exact."""
import ctypes as C

import numpy as np
import pytest

from pailliercryptolib_python_amd import _native
from tests._util import DevArray, host_ptr, ints_to_limbs, limbs_to_ints, rand_below

pytestmark = pytest.mark.gpu


def make_modulus(bits, seed):
    rng = np.random.default_rng(seed)
    m = int.from_bytes(rng.bytes((bits + 7) // 8), "little") & ((1 << bits) - 1)
    return m | (1 << (bits - 1)) | 1


class Modulus:
    def __init__(self, M):
        self.lib = _native.load()
        self.M = M
        self.w32 = (M.bit_length() + 31) // 32
        h = C.c_void_p()
        words = ints_to_limbs([M], self.w32)
        _native.check(self.lib.pai_modulus_create(host_ptr(words), self.w32, 0, C.byref(h)))
        self.h = h

    def __del__(self):
        try:
            self.lib.pai_modulus_destroy(self.h)
        except Exception:
            pass


# bit lengths: the BASELINE key sizes' moduli, plus the tightest modulus each geometry admits
# (29*NL - 2 bits: R is only 4..8 M there, so the lazy results regularly exceed M and exercise the
# final conditional subtraction and the cross-lane carry passes)
BITS = [1024, 1042, 2048, 2086, 3072, 3246, 4096, 4174, 6144, 8192, 8350, 515, 97]


@pytest.mark.parametrize("bits", BITS)
def test_modmul_matches_python(bits):
    M = make_modulus(bits, 100 + bits)
    mod = Modulus(M)
    rng = np.random.default_rng(bits)
    N = 777
    a = rand_below(rng, M, N)
    b = rand_below(rng, M, N)
    a[0], b[0] = M - 1, M - 1
    a[1], b[1] = 0, 5
    a[2], b[2] = 1, 1
    da, db = DevArray(ints_to_limbs(a, mod.w32)), DevArray(ints_to_limbs(b, mod.w32))
    out = DevArray(shape=(N, mod.w32))
    _native.check(mod.lib.pai_modmul(mod.h, da.ptr, db.ptr, 0, N, out.ptr, None))
    got = limbs_to_ints(out.get())
    assert got == [x * y % M for x, y in zip(a, b)]
    # broadcast of a single right operand
    _native.check(mod.lib.pai_modmul(mod.h, da.ptr, db.ptr, 1, N, out.ptr, None))
    assert limbs_to_ints(out.get()) == [x * b[0] % M for x in a]


@pytest.mark.parametrize("bits", [1024, 2048, 2086, 3072, 4096, 8192])
def test_modexp_fixed_matches_python(bits):
    M = make_modulus(bits, 200 + bits)
    mod = Modulus(M)
    rng = np.random.default_rng(bits + 1)
    N = 300 if bits <= 4096 else 70
    base = rand_below(rng, M, N)
    base[0], base[1], base[2] = 0, 1, M - 1
    for ebits in (1, 5, 64, 131):
        e = int.from_bytes(rng.bytes(ebits // 8 + 1), "little") % (1 << ebits) | (1 << (ebits - 1))
        ew = (ebits + 31) // 32
        he = ints_to_limbs([e], ew)
        db = DevArray(ints_to_limbs(base, mod.w32))
        out = DevArray(shape=(N, mod.w32))
        _native.check(mod.lib.pai_modexp_fixed(mod.h, db.ptr, host_ptr(he), ew, N, out.ptr, None))
        assert limbs_to_ints(out.get()) == [pow(x, e, M) for x in base], f"ebits={ebits}"


@pytest.mark.parametrize("bits", [1024, 2048, 4096])
def test_modexp_var_matches_python(bits):
    M = make_modulus(bits, 300 + bits)
    mod = Modulus(M)
    rng = np.random.default_rng(bits + 2)
    N = 333
    base = rand_below(rng, M, N)
    es = [int(x) for x in rng.integers(0, 1 << 53, size=N)]
    es[0], es[1], es[2], es[3] = 0, 1, 2, (1 << 53) - 1
    db = DevArray(ints_to_limbs(base, mod.w32))
    de = DevArray(ints_to_limbs(es, 2))
    out = DevArray(shape=(N, mod.w32))
    _native.check(mod.lib.pai_modexp_var(mod.h, db.ptr, 0, de.ptr, 2, 53, 0, N, out.ptr, None))
    assert limbs_to_ints(out.get()) == [pow(x, e, M) for x, e in zip(base, es)]
    # broadcast exponent, broadcast base
    _native.check(mod.lib.pai_modexp_var(mod.h, db.ptr, 0, de.ptr, 2, 53, 1, N, out.ptr, None))
    assert limbs_to_ints(out.get()) == [pow(x, es[0], M) for x in base]
    es2 = [int(x) for x in rng.integers(0, 1 << 20, size=N)]
    de2 = DevArray(ints_to_limbs(es2, 1))
    _native.check(mod.lib.pai_modexp_var(mod.h, db.ptr, 1, de2.ptr, 1, 20, 0, N, out.ptr, None))
    assert limbs_to_ints(out.get()) == [pow(base[0], e, M) for e in es2]
