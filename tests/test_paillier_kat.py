"""Known-answer vectors (tests/golden/paillier_kat.json, generated with bare CPython ``pow`` by
tests/golden/make_paillier_kat.py): the oracle must reproduce them on the CPU, the device library through the C ABI."""
import ctypes as C
import json
from pathlib import Path

import numpy as np
import pytest

from oracle import paillier_oracle as orc

KAT = json.loads((Path(__file__).parent / "golden" / "paillier_kat.json").read_text())


def _key(bits, rec, djn=True):
    return orc.make_key(int(rec["p"], 16), int(rec["q"], 16), djn_x=int(rec["djn_x"], 16) if djn else None, bits=bits)


@pytest.mark.parametrize("bits", sorted(int(b) for b in KAT))
def test_oracle_reproduces_the_known_answers(bits):
    rec = KAT[str(bits)]
    key = _key(bits, rec)
    std = _key(bits, rec, djn=False)
    assert key.hs == int(rec["hs"], 16) and key.randbits == rec["randbits"]
    for c in rec["cases"]:
        m, raw, ct = int(c["m"], 16), int(c["raw"], 16), int(c["ct_djn"], 16)
        assert orc.raw_encrypt(m, key.n) == raw
        assert orc.encrypt(key, m, int(c["r_djn"], 16)) == ct
        assert orc.encrypt(std, m, int(c["r_std"], 16)) == int(c["ct_std"], 16)
        assert orc.decrypt_crt(key, ct) == m == orc.decrypt_lambda(key, ct) == orc.decrypt_crt(std, int(c["ct_std"], 16))
        assert orc.ct_add(ct, int(c["other"], 16), key.nsq) == int(c["add"], 16)
        assert orc.ct_mul(ct, int(c["e"], 16), key.nsq) == int(c["mul"], 16)
        assert orc.ct_inv(ct, key.nsq) == int(c["inv"], 16)


@pytest.mark.gpu
@pytest.mark.parametrize("bits", sorted(int(b) for b in KAT))
def test_device_reproduces_the_known_answers(bits):
    from pailliercryptolib_python_amd import _native
    from tests._util import DevArray, ints_to_limbs, limbs_to_ints
    from tests.test_gpu_paillier_abi import NativeKey

    rec = KAT[str(bits)]
    cases = rec["cases"]
    N = len(cases)
    col = lambda k: [int(c[k], 16) for c in cases]          # noqa: E731
    for djn in (True, False):
        nk = NativeKey(_key(bits, rec, djn=djn))
        lib, key = nk.lib, nk.key
        dm = DevArray(ints_to_limbs(col("m"), nk.nw))
        ct = DevArray(shape=(N, nk.cw))
        _native.check(lib.pai_raw_encrypt(nk.pk, dm.ptr, N, ct.ptr, None))
        assert limbs_to_ints(ct.get()) == col("raw")
        dr = DevArray(ints_to_limbs(col("r_djn" if djn else "r_std"), nk.rw))
        _native.check(lib.pai_encrypt(nk.pk, dm.ptr, dr.ptr, N, ct.ptr, None))
        assert limbs_to_ints(ct.get()) == col("ct_djn" if djn else "ct_std")
        out = DevArray(shape=(N, nk.nw))
        _native.check(lib.pai_decrypt(nk.sk, ct.ptr, N, out.ptr, None))
        assert limbs_to_ints(out.get()) == col("m")
        if not djn:
            continue
        res = DevArray(shape=(N, nk.cw))
        do = DevArray(ints_to_limbs(col("other"), nk.cw))
        _native.check(lib.pai_ct_add(nk.pk, ct.ptr, do.ptr, 0, N, res.ptr, None))
        assert limbs_to_ints(res.get()) == col("add")
        de = DevArray(ints_to_limbs(col("e"), 2))
        _native.check(lib.pai_ct_mul(nk.pk, ct.ptr, de.ptr, 2, 53, 0, N, res.ptr, None))
        assert limbs_to_ints(res.get()) == col("mul")
        full = [key.n - 1 - i for i in range(N)]
        df = DevArray(ints_to_limbs(full, nk.nw))
        _native.check(lib.pai_ct_mul(nk.pk, ct.ptr, df.ptr, nk.nw, bits, 0, N, res.ptr, None))
        assert limbs_to_ints(res.get()) == col("mul_full")
        _native.check(lib.pai_ct_invert(nk.pk, ct.ptr, N, res.ptr, None))
        assert limbs_to_ints(res.get()) == col("inv")
        dd = DevArray(np.full(N, 7, dtype=np.int32))
        _native.check(lib.pai_ct_pow2(nk.pk, ct.ptr, dd.ptr, 0, N, None))
        assert limbs_to_ints(ct.get()) == col("pow2_7")
