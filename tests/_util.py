"""Test helpers: device arrays through the C ABI's own allocator (no torch needed at this level)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from pailliercryptolib_python_amd import _native


class DevArray:
    """A uint32/int32 matrix in device memory owned by the test."""

    def __init__(self, host: np.ndarray | None = None, shape=None, dtype=np.uint32, device: int = 0):
        self.device = device
        self.lib = _native.load()
        if host is not None:
            host = np.ascontiguousarray(host)
            shape, dtype = host.shape, host.dtype
        self.shape, self.dtype = tuple(shape), np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        p = C.c_void_p()
        _native.check(self.lib.pai_malloc(device, max(self.nbytes, 4), C.byref(p)))
        self.ptr = p
        if host is not None and self.nbytes:
            _native.check(self.lib.pai_memcpy_h2d(device, self.ptr, host.ctypes.data_as(C.c_void_p), self.nbytes, None))

    def get(self) -> np.ndarray:
        out = np.empty(self.shape, dtype=self.dtype)
        if self.nbytes:
            _native.check(self.lib.pai_memcpy_d2h(self.device, out.ctypes.data_as(C.c_void_p), self.ptr, self.nbytes, None))
        return out

    def free(self):
        if self.ptr:
            self.lib.pai_free(self.device, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def ints_to_limbs(vals, L):
    out = np.zeros((len(vals), L), dtype=np.uint32)
    for i, v in enumerate(vals):
        out[i] = np.frombuffer(int(v).to_bytes(4 * L, "little"), dtype="<u4")
    return out


def limbs_to_ints(arr):
    arr = np.ascontiguousarray(arr, dtype="<u4")
    return [int.from_bytes(row.tobytes(), "little") for row in arr]


def rand_below(rng, bound: int, count: int):
    nbytes = (bound.bit_length() + 7) // 8 + 8
    return [int.from_bytes(rng.bytes(nbytes), "little") % bound for _ in range(count)]


def host_ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def tune(monkeypatch, name: str, value) -> None:
    """Adds / replaces `name=value` in PAI_TUNE (the library's list of test and tuning parameters) for
    the current test; value None removes the entry."""
    import os

    cur = [kv for kv in os.environ.get("PAI_TUNE", "").split(",") if kv and kv.split("=")[0] != name]
    if value is not None:
        cur.append(f"{name}={value}")
    monkeypatch.setenv("PAI_TUNE", ",".join(cur))


def disable(monkeypatch, name: str, on: bool = True) -> None:
    """Adds (on) / removes `name` in PAI_DISABLE (engines and forms the library is to leave out) for the current test."""
    import os

    cur = [k for k in os.environ.get("PAI_DISABLE", "").split(",") if k and k != name]
    if on:
        cur.append(name)
    monkeypatch.setenv("PAI_DISABLE", ",".join(cur))


def pow_many(bases, exps, M: int, check: int = 3):
    """[pow(b, e, M) for b, e in zip(bases, exps)] (exps: a list, or one int for all) through the C oracle's plain Montgomery
    exponentiation on the host threads (oracle/paillier_ref.c: orc_modexp_batch), spot-checked against CPython's pow on
    `check` elements spread over the batch — the expectations of wide-key tests, where CPython's pow costs 0.1-0.3 s per
    element, without weakening them (the C restatement is itself held to CPython in tests/test_oracle.py)."""
    from oracle import c_oracle as co

    bases = [int(b) % M for b in bases]
    n = len(bases)
    if n == 0:
        return []
    if isinstance(exps, int):
        exps = [exps] * n
    exps = [int(e) for e in exps]
    L = (M.bit_length() + 63) // 64
    if not (M & 1) or L > 136:
        return [pow(b, e, M) for b, e in zip(bases, exps)]
    nm, n0, r2 = co._mont_consts(M, L)
    b64 = co._as_u64_rows(ints_to_limbs(bases, 2 * L), L)
    ebits = max(max(e.bit_length() for e in exps), 1)
    stride = (ebits + 63) // 64
    ev = np.concatenate([co._u64(e, stride) for e in exps])
    out = np.zeros_like(b64)
    rc = co.lib().orc_modexp_batch(n, L, co._p(nm), n0, co._p(r2), co._p(b64), co._p(ev), stride, ebits, co._p(out), 0)
    assert rc == 0
    got = limbs_to_ints(out.view(np.uint32))
    for i in sorted({int(v) for v in np.linspace(0, n - 1, min(check, n))}):
        assert got[i] == pow(bases[i], exps[i], M), "C oracle and CPython pow disagree"
    return got


def djn_encrypt_many(key, ms, rs):
    """[orc.encrypt(key, m, r)] for a DJN key: (1 + m n) hs^r mod n^2 with the obfuscators through pow_many."""
    obf = pow_many([key.hs] * len(ms), rs, key.nsq)
    return [(1 + m * key.n) % key.nsq * o % key.nsq for m, o in zip(ms, obf)]


def djn_obfuscate_many(key, cts, rs):
    """[orc.apply_obfuscator(key, c, r)]: c hs^r mod n^2."""
    obf = pow_many([key.hs] * len(cts), rs, key.nsq)
    return [c * o % key.nsq for c, o in zip(cts, obf)]
