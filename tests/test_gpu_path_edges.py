"""GPU: both sides of every batch-size switch of the C API (round 6; VERDICT r05 "next" #7).

pai_path_edges lists, for a key, the batch sizes at which an operation changes kernel family (csrc/path_ranges.hpp).  For every
edge E of every operation the call runs at N = E and N = E + 1 on the default dispatch and must give, bit for bit, what the
throughput kernels alone give (PAI_LATENCY_MAX=0 switches every small-batch and mid-size path off), with oracle samples on top.
The differential fuzz (tools/fuzz_gpu.py) crosses the switches at random sizes; this pins them at their exact positions."""
import ctypes as C

import numpy as np
import pytest

from oracle import paillier_oracle as orc
from pailliercryptolib_python_amd import _native
from tests._util import DevArray, ints_to_limbs, limbs_to_ints, pow_many
from tests.test_gpu_paillier_abi import NativeKey, bench_key, seeded_key

pytestmark = pytest.mark.gpu


def edges_of(nk, op):
    cnt = C.c_int(0)
    buf = (C.c_size_t * 16)()
    _native.check(nk.lib.pai_path_edges(nk.pk, op, buf, 16, C.byref(cnt)))
    return [int(buf[i]) for i in range(min(cnt.value, 16))]


@pytest.fixture(scope="module", params=[2048, 1024])
def pool(request):
    """One key and one resident pool of (m, r, ciphertext) rows as large as the largest edge needs."""
    bits = request.param
    nk = NativeKey(bench_key() if bits == 2048 else seeded_key(bits))
    key = nk.key
    sizes = [e + 1 for op in range(4) for e in edges_of(nk, op)]
    N = max(sizes)
    rng = np.random.default_rng(31 + bits)
    m = rng.integers(0, 1 << 32, (N, nk.nw), dtype=np.uint64).astype(np.uint32)
    m[:, -1] &= 0x0FFFFFFF                                   # < n
    r = rng.integers(0, 1 << 32, (N, nk.rw), dtype=np.uint64).astype(np.uint32)
    top = key.randbits - 32 * (nk.rw - 1)
    if top < 32:
        r[:, -1] &= (1 << top) - 1
    dm, dr, dct = DevArray(m), DevArray(r), DevArray(shape=(N, nk.cw))
    _native.check(nk.lib.pai_encrypt(nk.pk, dm.ptr, dr.ptr, N, dct.ptr, None))
    return nk, N, m, r, dm, dr, dct


def both_ways(monkeypatch, call):
    """call() on the default dispatch and with every small-batch / mid-size path off."""
    got = call()
    monkeypatch.setenv("PAI_LATENCY_MAX", "0")
    ref = call()
    monkeypatch.delenv("PAI_LATENCY_MAX")
    return got, ref


def test_pai_path_edges_lists_ascending_switch_points(pool, monkeypatch):
    nk = pool[0]
    for op in range(4):
        e = edges_of(nk, op)
        assert e == sorted(set(e)) and all(v > 0 for v in e) and len(e) >= 2, (op, e)
    base = edges_of(nk, 0)
    monkeypatch.setenv("PAI_LATENCY_MAX", "0")               # no small-batch switch left, the mid-size range gone as well
    off = edges_of(nk, 0)
    monkeypatch.delenv("PAI_LATENCY_MAX")
    assert len(off) < len(base)


def test_encrypt_on_both_sides_of_every_switch(pool, monkeypatch):
    nk, N, m, r, dm, dr, dct = pool
    key = nk.key
    full = dct.get()
    for e in edges_of(nk, 1):
        for n in (e, e + 1):
            out = DevArray(shape=(n, nk.cw))
            got, ref = both_ways(monkeypatch, lambda: (_native.check(nk.lib.pai_encrypt(nk.pk, dm.ptr, dr.ptr, n, out.ptr, None)), out.get())[1])
            assert np.array_equal(got, ref) and np.array_equal(got, full[:n]), (key.bits, n)
    idx = [0, 1, N // 2, N - 1]
    assert limbs_to_ints(full[idx]) == [orc.encrypt(key, a, b) for a, b in zip(limbs_to_ints(m[idx]), limbs_to_ints(r[idx]))]


def test_decrypt_on_both_sides_of_every_switch(pool, monkeypatch):
    nk, N, m, r, dm, dr, dct = pool
    for e in edges_of(nk, 0):
        for n in (e, e + 1):
            out = DevArray(shape=(n, nk.nw))
            got, ref = both_ways(monkeypatch, lambda: (_native.check(nk.lib.pai_decrypt(nk.sk, dct.ptr, n, out.ptr, None)), out.get())[1])
            assert np.array_equal(got, ref) and np.array_equal(got, m[:n]), (nk.key.bits, n)


def test_ct_mul_on_both_sides_of_every_switch(pool, monkeypatch):
    nk, N, m, r, dm, dr, dct = pool
    key = nk.key
    rng = np.random.default_rng(77)
    es = [int(v) | 1 << 52 for v in rng.integers(0, 1 << 52, N)]
    de = DevArray(ints_to_limbs(es, 2))
    cts = None
    for e in edges_of(nk, 2):
        for n in (e, e + 1):
            out = DevArray(shape=(n, nk.cw))
            got, ref = both_ways(monkeypatch, lambda: (_native.check(nk.lib.pai_ct_mul(nk.pk, dct.ptr, de.ptr, 2, 53, 0, n, out.ptr, None)), out.get())[1])
            assert np.array_equal(got, ref), (key.bits, n)
            cts = got
    idx = [0, 1, cts.shape[0] // 2, cts.shape[0] - 1]
    full = limbs_to_ints(dct.get()[idx])
    assert limbs_to_ints(cts[idx]) == pow_many(full, [es[i] for i in idx], key.nsq)


def test_ct_add_forms_on_both_sides_of_every_switch(pool, monkeypatch):
    nk, N, m, r, dm, dr, dct = pool
    key = nk.key
    rng = np.random.default_rng(78)
    for e in edges_of(nk, 3):
        for n in (e, e + 1):
            if n + 1 > N:
                continue
            a_ptr = dct.ptr
            b = DevArray(dct.get()[1:n + 1])                   # the neighbour's ciphertext
            delta = rng.integers(-2, 3, n).astype(np.int32)
            dd = DevArray(delta)
            o1, o2, o3 = DevArray(shape=(n, nk.cw)), DevArray(shape=(n, nk.cw)), DevArray(shape=(n, nk.cw))

            def call():
                _native.check(nk.lib.pai_ct_add(nk.pk, a_ptr, b.ptr, 0, n, o1.ptr, None))
                _native.check(nk.lib.pai_ct_add_aligned(nk.pk, a_ptr, b.ptr, 0, dd.ptr, n, o2.ptr, None))
                _native.check(nk.lib.pai_ct_mont_mul(nk.pk, a_ptr, b.ptr, 0, n, o3.ptr, None))
                return o1.get(), o2.get(), o3.get()

            monkeypatch.setenv("PAI_LAT_ADD_MAX", "0")
            ref = call()
            monkeypatch.delenv("PAI_LAT_ADD_MAX")
            got = call()
            assert all(np.array_equal(g, w) for g, w in zip(got, ref)), (key.bits, n)
            idx = [0, n // 2, n - 1]
            A, B = limbs_to_ints(dct.get()[idx]), limbs_to_ints(b.get()[idx])
            assert limbs_to_ints(got[0][idx]) == [x * y % key.nsq for x, y in zip(A, B)]
            assert limbs_to_ints(got[1][idx]) == [pow(x, 1 << max(0, -int(d)), key.nsq) * pow(y, 1 << max(0, int(d)), key.nsq) % key.nsq
                                                  for x, y, d in zip(A, B, delta[idx])]
