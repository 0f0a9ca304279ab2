"""ctypes front end of the plain-C oracle ``oracle/paillier_ref.c`` (TEST INFRASTRUCTURE ONLY).

Used by tests for batches too large for CPython ``pow`` and by ``bench.py`` as the ``cpu_baseline``
("kind": "port": the reference's own CPU path — ipcl + IPP-Crypto — is not in /root/reference and
cannot be built here).  Constants that need division are computed here with Python ints.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path
from typing import Optional

import numpy as np

from . import paillier_oracle as orc

HERE = Path(__file__).resolve().parent
SRC = HERE / "paillier_ref.c"
SRC_IFMA = HERE / "paillier_ifma.c"      # included by paillier_ref.c when the host compiler targets AVX512-IFMA
LIB = HERE / "_build" / "libpaillier_oracle.so"


def build(force: bool = False) -> Path:
    if LIB.exists() and not force and LIB.stat().st_mtime >= max(SRC.stat().st_mtime, SRC_IFMA.stat().st_mtime):
        return LIB
    LIB.parent.mkdir(parents=True, exist_ok=True)
    cmd = ["gcc", "-O3", "-march=native", "-fopenmp", "-shared", "-fPIC", str(SRC), "-o", str(LIB), "-ldl"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("gcc failed:\n" + res.stderr)
    return LIB


_lib = None


class _PrimeCtx(C.Structure):
    _fields_ = [
        ("s2", C.c_void_p), ("s2_r2", C.c_void_p), ("s2_0inv", C.c_uint64),
        ("e", C.c_void_p), ("ebits", C.c_int),
        ("s", C.c_void_p), ("s_r2", C.c_void_p), ("s_0inv", C.c_uint64),
        ("sinv2", C.c_void_p), ("hR", C.c_void_p),
    ]


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(str(build()))
        _lib.orc_max_threads.restype = C.c_int
    return _lib


def gmp_available() -> bool:
    lib().orc_gmp_available.restype = C.c_int
    return bool(lib().orc_gmp_available())


def ifma_available() -> bool:
    """True when the library was compiled with the AVX512-IFMA mb8 kernels and this CPU has the instructions."""
    lib().orc_ifma_available.restype = C.c_int
    return bool(lib().orc_ifma_available())


class _IfmaMod(C.Structure):
    _fields_ = [("bits", C.c_int), ("r2_52", C.c_void_p), ("k0", C.c_uint64)]


def _ifma_consts(M: int):
    """(bits, R^2 mod M as 52-bit limbs, -M^-1 mod 2^52) for R = 2^(52 L), 52 L >= bits + 2."""
    bits = M.bit_length()
    L = (bits + 2 + 51) // 52
    R = 1 << (52 * L)
    r2 = R * R % M
    limbs = np.array([(r2 >> (52 * i)) & ((1 << 52) - 1) for i in range(L)], dtype=np.uint64)
    return bits, limbs, (-pow(M, -1, 1 << 52)) % (1 << 52)


def ifma_modexp(M: int, base32: np.ndarray, e, threads: int = 0) -> np.ndarray:
    """out[i] = base[i]^e mod M through the mb8 kernel; e an int (shared) or a list of ints (per element)."""
    L = (M.bit_length() + 63) // 64
    bits, r2, k0 = _ifma_consts(M)
    b = _as_u64_rows(base32, L)
    if isinstance(e, int):
        ebits = max(e.bit_length(), 1)
        ev, stride = _u64(e, (ebits + 63) // 64), 0
    else:
        ebits = max(max(int(v).bit_length() for v in e), 1)
        stride = (ebits + 63) // 64
        ev = np.concatenate([_u64(int(v), stride) for v in e])
    out = np.zeros_like(b)
    rc = lib().orc_ifma_modexp_batch(b.shape[0], bits, L, _p(_u64(M, L)), _p(r2), C.c_uint64(k0), _p(b), _p(ev), stride, ebits,
                                     _p(out), threads)
    assert rc == 0
    return out.view(np.uint32)[:, : base32.shape[1]]


def max_threads() -> int:
    """Host threads worth using: min(OpenMP default, CPU affinity, cgroup CPU quota)."""
    n = int(lib().orc_max_threads())
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if quota != "max":
            n = min(n, max(1, int(round(int(quota) / int(period)))))
    except Exception:
        pass
    return max(1, n)


def _u64(v: int, L: int) -> np.ndarray:
    return np.frombuffer(int(v).to_bytes(8 * L, "little"), dtype="<u8").copy()


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def _mont_consts(M: int, L: int):
    R = 1 << (64 * L)
    return _u64(M, L), C.c_uint64((-pow(M, -1, 1 << 64)) % (1 << 64)), _u64(R * R % M, L)


def _as_u64_rows(a32: np.ndarray, L64: int) -> np.ndarray:
    a32 = np.ascontiguousarray(a32, dtype=np.uint32)
    N, W = a32.shape
    if W < 2 * L64:
        a32 = np.concatenate([a32, np.zeros((N, 2 * L64 - W), dtype=np.uint32)], axis=1)
    assert a32.shape[1] == 2 * L64
    return np.ascontiguousarray(a32).view(np.uint64)


def modexp(M: int, base32: np.ndarray, e: int, threads: int = 0) -> np.ndarray:
    """out[i] = base[i]^e mod M on [N][W] u32 limb matrices."""
    L = (M.bit_length() + 63) // 64
    n, n0, r2 = _mont_consts(M, L)
    b = _as_u64_rows(base32, L)
    ebits = max(e.bit_length(), 1)
    ev = _u64(e, (ebits + 63) // 64)
    out = np.zeros_like(b)
    rc = lib().orc_modexp_batch(b.shape[0], L, _p(n), n0, _p(r2), _p(b), _p(ev), 0, ebits, _p(out), threads)
    assert rc == 0
    return out.view(np.uint32)[:, : base32.shape[1]]


def modmul(M: int, a32: np.ndarray, b32: np.ndarray, threads: int = 0) -> np.ndarray:
    L = (M.bit_length() + 63) // 64
    n, n0, r2 = _mont_consts(M, L)
    a, b = _as_u64_rows(a32, L), _as_u64_rows(b32, L)
    out = np.zeros_like(a)
    rc = lib().orc_modmul_batch(a.shape[0], L, _p(n), n0, _p(r2), _p(a), _p(b), _p(out), threads)
    assert rc == 0
    return out.view(np.uint32)[:, : a32.shape[1]]


class COracleKey:
    """Pre-computed constants of one key for the C oracle."""

    def __init__(self, key: orc.OracleKey):
        self.key = key
        self.Ln = (key.bits + 63) // 64
        self.Lh = (max(key.p.bit_length(), key.q.bit_length()) + 63) // 64
        assert 2 * self.Lh == self.Ln or 2 * self.Lh >= self.Ln
        self.n = _u64(key.n, self.Ln)
        self.nsq, self.nsq0, self.nsq_r2 = _mont_consts(key.nsq, 2 * self.Ln)
        self.hs = _u64(key.hs, 2 * self.Ln) if key.hs is not None else None
        k = orc.crt_constants(key)
        self._keep = []
        self.pc = self._prime(key.p, k["hp"])
        self.qc = self._prime(key.q, k["hq"])
        Rq = 1 << (64 * self.Lh)
        self.pinvqR = _u64(k["pinv_q"] * Rq % key.q, self.Lh)

    def _prime(self, s: int, h: int) -> _PrimeCtx:
        Ln, Lh = self.Ln, self.Lh
        s2n, s20, s2r2 = _mont_consts(s * s, Ln)
        sn, s0, sr2 = _mont_consts(s, Lh)
        e = s - 1
        ev = _u64(e, (e.bit_length() + 63) // 64)
        sinv2 = _u64(pow(s, -1, 1 << (64 * Lh)), Lh)
        hR = _u64(h * (1 << (64 * Lh)) % s, Lh)
        self._keep += [s2n, s2r2, sn, sr2, ev, sinv2, hR]
        return _PrimeCtx(_p(s2n), _p(s2r2), s20, _p(ev), e.bit_length(), _p(sn), _p(sr2), s0, _p(sinv2), _p(hR))

    def encrypt_djn(self, m32: np.ndarray, r32: np.ndarray, threads: int = 0) -> np.ndarray:
        key = self.key
        m = _as_u64_rows(m32, self.Ln)
        Lr = (key.randbits + 63) // 64
        r = _as_u64_rows(r32, Lr)
        N = m.shape[0]
        ct = np.zeros((N, 2 * self.Ln), dtype=np.uint64)
        rc = lib().orc_encrypt_djn_batch(N, self.Ln, _p(self.n), _p(self.nsq), self.nsq0, _p(self.nsq_r2), _p(self.hs),
                                         _p(m), _p(r), Lr, key.randbits, _p(ct), threads)
        assert rc == 0
        return ct.view(np.uint32)

    def decrypt_crt(self, ct32: np.ndarray, threads: int = 0) -> np.ndarray:
        ct = _as_u64_rows(ct32, 2 * self.Ln)
        N = ct.shape[0]
        m = np.zeros((N, self.Ln), dtype=np.uint64)
        rc = lib().orc_decrypt_crt_batch(N, self.Ln, self.Lh, C.byref(self.pc), C.byref(self.qc), _p(self.pinvqR),
                                         _p(ct), _p(m), threads)
        assert rc == 0
        return m.view(np.uint32)

    # ---- the same two operations through libgmp (dlopen'ed by the C side) when it is installed ----
    def gmp_encrypt_djn(self, m32: np.ndarray, r32: np.ndarray, threads: int = 0) -> np.ndarray:
        key = self.key
        m = _as_u64_rows(m32, self.Ln)
        Lr = (key.randbits + 63) // 64
        r = _as_u64_rows(r32, Lr)
        N = m.shape[0]
        ct = np.zeros((N, 2 * self.Ln), dtype=np.uint64)
        rc = lib().orc_gmp_encrypt_djn_batch(N, self.Ln, _p(self.n), _p(self.nsq), _p(self.hs), _p(m), _p(r), Lr,
                                             _p(ct), threads)
        assert rc == 0, "libgmp not available"
        return ct.view(np.uint32)

    def gmp_decrypt_crt(self, ct32: np.ndarray, threads: int = 0) -> np.ndarray:
        key = self.key
        k = orc.crt_constants(key)
        ct = _as_u64_rows(ct32, 2 * self.Ln)
        N = ct.shape[0]
        m = np.zeros((N, self.Ln), dtype=np.uint64)
        args = [_u64(v, self.Lh) for v in (key.p, key.q, k["hp"], k["hq"], k["pinv_q"])]
        rc = lib().orc_gmp_decrypt_crt_batch(N, self.Ln, self.Lh, *[_p(a) for a in args], _p(ct), _p(m), threads)
        assert rc == 0, "libgmp not available"
        return m.view(np.uint32)

    # ---- the same two operations on the AVX512-IFMA mb8 kernels (oracle/paillier_ifma.c) ----
    def _ifma_mod(self, M: int) -> _IfmaMod:
        bits, r2, k0 = _ifma_consts(M)
        self._keep.append(r2)
        return _IfmaMod(bits, _p(r2), k0)

    def ifma_encrypt_djn(self, m32: np.ndarray, r32: np.ndarray, threads: int = 0) -> np.ndarray:
        key = self.key
        if not hasattr(self, "_im_nsq"):
            self._im_nsq = self._ifma_mod(key.nsq)
        m = _as_u64_rows(m32, self.Ln)
        Lr = (key.randbits + 63) // 64
        r = _as_u64_rows(r32, Lr)
        N = m.shape[0]
        ct = np.zeros((N, 2 * self.Ln), dtype=np.uint64)
        rc = lib().orc_ifma_encrypt_djn_batch(N, self.Ln, _p(self.n), _p(self.nsq), self.nsq0, _p(self.nsq_r2), _p(self.hs),
                                              _p(m), _p(r), Lr, key.randbits, _p(ct), C.byref(self._im_nsq), threads)
        assert rc == 0
        return ct.view(np.uint32)

    def ifma_decrypt_crt(self, ct32: np.ndarray, threads: int = 0) -> np.ndarray:
        key = self.key
        if not hasattr(self, "_im_p"):
            self._im_p, self._im_q = self._ifma_mod(key.p * key.p), self._ifma_mod(key.q * key.q)
        ct = _as_u64_rows(ct32, 2 * self.Ln)
        N = ct.shape[0]
        m = np.zeros((N, self.Ln), dtype=np.uint64)
        rc = lib().orc_ifma_decrypt_crt_batch(N, self.Ln, self.Lh, C.byref(self.pc), C.byref(self.qc), _p(self.pinvqR),
                                              _p(ct), _p(m), C.byref(self._im_p), C.byref(self._im_q), threads)
        assert rc == 0
        return m.view(np.uint32)


class CApi:
    """The reference's API-level compositions (src/ipcl_python/ipcl_python.py) driven on the C port's batch primitives:
    what the reference's Python layer does around its native calls, with this file's kernels in the place of
    ipcl / IPP-Crypto.  Used by bench.py's `reference_bench` CPU leg (bench/bench_ipcl_python.py:22-78) and checked
    against the Python-int oracle in tests/test_oracle.py.  Ciphertexts are [N][2 k/32] uint32 rows, exponents lists."""

    def __init__(self, key: orc.OracleKey, threads: int = 0):
        self.key, self.ck, self.threads = key, COracleKey(key), threads
        self.ifma = ifma_available()
        self.n_words = (key.bits + 31) // 32

    def _encode(self, values):
        """fixedpoint.py:54-96 per element (the reference's own per-element Python loop, ipcl_python.py:135-141)."""
        encs, expos = [], []
        for v in values:
            e, x = orc.fp_encode(v, self.key.n, self.key.max_int)
            encs.append(e), expos.append(x)
        return encs, expos

    def _pow(self, ct32: np.ndarray, es) -> np.ndarray:
        if self.ifma:
            return ifma_modexp(self.key.nsq, ct32, [int(e) for e in es], threads=self.threads).view(np.uint32)
        nsq = self.key.nsq
        return orc.ints_to_limbs([pow(c, int(e), nsq) for c, e in zip(orc.limbs_to_ints(ct32), es)], 2 * self.n_words)

    def encrypt(self, values, r32: Optional[np.ndarray] = None):
        """ipcl_python.py:108-147 (DJN)."""
        encs, expos = self._encode(values)
        m32 = orc.ints_to_limbs(encs, self.n_words)
        if r32 is None:
            rw = (self.key.randbits + 31) // 32
            r32 = np.frombuffer(os.urandom(4 * rw * len(encs)), dtype=np.uint32).reshape(len(encs), rw).copy()
            if self.key.randbits % 32:
                r32[:, -1] &= (1 << (self.key.randbits % 32)) - 1
        enc = self.ck.ifma_encrypt_djn if self.ifma else self.ck.encrypt_djn
        return np.ascontiguousarray(enc(m32, r32, threads=self.threads)), expos

    def raw_encrypt(self, values):
        """ipcl_python.py:103-106: 1 + m n (no obfuscator)."""
        encs, expos = self._encode(values)
        n = self.key.n
        return orc.ints_to_limbs([1 + m * n for m in encs], 2 * self.n_words), expos

    def decrypt(self, ct32: np.ndarray, expos):
        """ipcl_python.py:219-245."""
        dec = self.ck.ifma_decrypt_crt if self.ifma else self.ck.decrypt_crt
        ms = orc.limbs_to_ints(dec(ct32, threads=self.threads))
        return [orc.fp_decode(m, e, self.key.n, self.key.max_int) for m, e in zip(ms, expos)]

    def _align(self, a32, ea, b32, eb):
        """ipcl_python.py:570-741: the lower-exponent side is raised by ct^(2^delta)."""
        if b32.shape[0] == 1 and a32.shape[0] > 1:
            b32, eb = np.repeat(b32, a32.shape[0], axis=0), list(eb) * a32.shape[0]
        ea_, eb_ = np.asarray(ea), np.asarray(eb)
        a32, b32 = a32.copy(), b32.copy()
        lo_b, lo_a = np.nonzero(ea_ > eb_)[0], np.nonzero(ea_ < eb_)[0]
        if len(lo_b):
            b32[lo_b] = self._pow(b32[lo_b], [1 << int(d) for d in (ea_ - eb_)[lo_b]])
        if len(lo_a):
            a32[lo_a] = self._pow(a32[lo_a], [1 << int(d) for d in (eb_ - ea_)[lo_a]])
        return a32, b32, [int(v) for v in np.maximum(ea_, eb_)]

    def add_ctct(self, a32, ea, b32, eb):
        """__raw_add on two ciphertext operands (ipcl_python.py:490-526)."""
        a, b, e = self._align(a32, ea, b32, eb)
        return modmul(self.key.nsq, a, b, threads=self.threads).view(np.uint32), e

    def add_ctpt(self, a32, ea, values):
        """ct + plaintext array: raw-encrypt, then add (ipcl_python.py:495-504)."""
        b32, eb = self.raw_encrypt(values)
        return self.add_ctct(a32, ea, b32, eb)

    def mul_ctpt(self, a32, ea, values):
        """ipcl_python.py:412-488: ct^mantissa, negative multipliers through the inverted ciphertext."""
        encs, expos = self._encode(values)
        n, nsq = self.key.n, self.key.nsq
        cond = n - self.key.max_int
        es, base = [], a32.copy()
        for i, pt in enumerate(encs):
            if pt >= cond:
                base[i] = orc.ints_to_limbs([pow(orc.limbs_to_ints(a32[i:i + 1])[0], -1, nsq)], 2 * self.n_words)[0]
                es.append(n - pt)
            else:
                es.append(pt)
        return self._pow(base, es), [a + b for a, b in zip(ea, expos)]
