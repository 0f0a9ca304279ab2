"""Host-side wire formats of the drop-in boundary (no GPU): BNUtils (ipcl_python.py:933-977) and the bytes <-> big number
converters pyByte2BN / BN2bytes (bindings/ipcl_bindings.cpp:100-138) as restated by bindings.ipclBigNumber, plus the
bulk limb packing the C ABI uses instead of one Python object per element (engine.ints_to_words / words_to_ints)."""
import numpy as np
import pytest

from pailliercryptolib_python_amd import BNUtils, engine
from pailliercryptolib_python_amd.bindings import ipclBigNumber

VALUES = [0, 1, 2, 255, 256, 2**32 - 1, 2**32, 2**64 + 5, (1 << 2047) + 12345, (1 << 4096) - 1]


@pytest.mark.parametrize("v", VALUES)
def test_bnutils_round_trip_and_minimal_little_endian_bytes(v):
    b = BNUtils.int2Bytes(v)
    assert b == v.to_bytes((v.bit_length() + 7) // 8, "little")            # ipcl_python.py:936-937: minimal length, 0 -> b""
    assert BNUtils.bytes2Int(b) == v
    bn = BNUtils.int2BN(v)
    assert BNUtils.BN2int(bn) == v and bn == v
    # BN2bytes pads to whole 32-bit words (ipcl_bindings.cpp:125,134); zero is one zero word
    wire = bn.to_bytes()
    assert len(wire) % 4 == 0 and len(wire) == 4 * max(1, (v.bit_length() + 31) // 32)
    assert int.from_bytes(wire, "little") == v
    # pyByte2BN accepts any length: a tail shorter than a word is zero-extended (ipcl_bindings.cpp:108-116)
    assert ipclBigNumber(b) == v and ipclBigNumber(b + b"\x00\x00\x00") == v


def test_static_constants_are_the_reference_special_cases():
    assert BNUtils.int2BN(0) is ipclBigNumber.Zero and BNUtils.int2BN(1) is ipclBigNumber.One and BNUtils.int2BN(2) is ipclBigNumber.Two
    assert int(ipclBigNumber.Zero) == 0 and int(ipclBigNumber.One) == 1 and int(ipclBigNumber.Two) == 2


def test_bulk_limb_matrix_is_little_endian_words_of_little_endian_limbs():
    vals = VALUES[:-1]
    L = 130
    w = engine.ints_to_words(vals, L)
    assert w.dtype == np.uint32 and w.shape == (len(vals), L)
    assert engine.words_to_ints(w) == vals
    # row i is exactly the zero-padded BN2bytes form of element i
    for row, v in zip(w, vals):
        assert row.tobytes() == v.to_bytes(4 * L, "little")
    with pytest.raises(OverflowError):
        engine.ints_to_words([1 << (32 * L)], L)


def test_bignumber_word_accessors():
    """ipclBigNumber.DwordSize / BitSize / __getitem__ / data and the list constructor
    (bindings/ipcl_bindings_classes.cpp:386-393, 422-432, 458-471): 32-bit words, little-endian."""
    import pytest

    v = (0xDEADBEEF << 64) | (0x12345678 << 32) | 0x9ABCDEF0
    b = ipclBigNumber([0x9ABCDEF0, 0x12345678, 0xDEADBEEF])
    assert int(b) == v and b == ipclBigNumber(v)
    assert b.DwordSize() == 3 and b.BitSize() == 96
    assert [b[i] for i in range(3)] == [0x9ABCDEF0, 0x12345678, 0xDEADBEEF]
    assert b.data() == (3, [0x9ABCDEF0, 0x12345678, 0xDEADBEEF])
    with pytest.raises(IndexError):
        b[3]
    assert ipclBigNumber.Zero.DwordSize() == 1 and ipclBigNumber.Zero[0] == 0
    with pytest.raises(TypeError):
        ipclBigNumber([1 << 32])
