"""Device-side engine objects: thin, typed wrappers over the C ABI working on torch tensors.

These take the place of the pybind11 classes ``ipclPublicKey`` / ``ipclPrivateKey`` /
``ipclPlainText`` / ``ipclCipherText`` (``bindings/ipcl_bindings_classes.cpp:12-378``).  Where the
reference moves ``std::vector<BigNumber>`` copies through Python lists on every call, a batch here is
ONE device tensor of little-endian 32-bit limbs, shape ``[N, words]``, dtype ``torch.int32`` (bit
pattern of uint32), and it stays in HBM between operations.  torch is used only for allocation,
streams and (in ``sharding.py``) ``torch.distributed``; all arithmetic is in ``libpaillier_hip.so``.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
import weakref
from typing import Optional

import numpy as np
import torch

from . import _native


def _require_cuda(device: torch.device) -> torch.device:
    if not torch.cuda.is_available():
        raise RuntimeError(
            "pailliercryptolib_python_amd needs an AMD GPU (gfx950): torch.cuda.is_available() is False "
            "and there is no CPU fallback"
        )
    return device


def int_to_words(v: int, words: int) -> np.ndarray:
    return np.frombuffer(int(v).to_bytes(4 * words, "little"), dtype="<u4").copy()


def ints_to_words(vals, words: int) -> np.ndarray:
    """list of non-negative Python ints -> [N][words] uint32 (little-endian limbs)."""
    n = len(vals)
    buf = bytearray(4 * words * n)
    step = 4 * words
    for i, v in enumerate(vals):
        buf[i * step:(i + 1) * step] = int(v).to_bytes(step, "little")
    return np.frombuffer(bytes(buf), dtype="<u4").reshape(n, words).copy()


def words_to_ints(arr: np.ndarray):
    arr = np.ascontiguousarray(arr, dtype="<u4")
    if arr.ndim == 1:
        arr = arr[None, :]
    raw = arr.tobytes()
    step = 4 * arr.shape[1]
    return [int.from_bytes(raw[i * step:(i + 1) * step], "little") for i in range(arr.shape[0])]


def to_device_words(host: np.ndarray, device: torch.device) -> torch.Tensor:
    """[N][W] uint32 numpy -> int32 torch tensor on `device` (same bits)."""
    host = np.ascontiguousarray(host, dtype=np.uint32)
    return torch.from_numpy(host.view(np.int32)).to(device, non_blocking=False)


def to_host_words(t: torch.Tensor) -> np.ndarray:
    return t.detach().cpu().contiguous().numpy().view(np.uint32)


def rows_slice(t: torch.Tensor, start: int, count: int, step: int = 1) -> torch.Tensor:
    """out[i] = t[start + i*step] on the device (pai_buf_slice: the containers' slice __getitem__, classes.cpp:224-262,328-366)."""
    t = t.contiguous()
    out = torch.empty((count, t.shape[1]), dtype=t.dtype, device=t.device)
    if count:
        _native.check(_native.load().pai_buf_slice(t.device.index or 0, _ptr(t), t.shape[1], int(start), int(count), int(step),
                                                   _ptr(out), _stream(t.device)))
    return out


def rows_rotate(t: torch.Tensor, shift: int) -> torch.Tensor:
    """out[i] = t[(i + shift) mod N] on the device (pai_buf_rotate: CipherText::rotate / PlainText::rotate)."""
    t = t.contiguous()
    out = torch.empty_like(t)
    if t.shape[0]:
        _native.check(_native.load().pai_buf_rotate(t.device.index or 0, _ptr(t), t.shape[1], t.shape[0], int(shift), _ptr(out),
                                                    _stream(t.device)))
    return out


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream(device: torch.device):
    if _raw_stream is not None and device.index is not None:
        return C.c_void_p(_raw_stream(device.index))             # the current stream's handle without building a Stream object
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


# ---- small host operands (pai_host_stage): exponents, shifts and codec inputs of small batches are read by the kernel from a
# pinned slot instead of crossing PCIe in a copy of their own (12 us per torch .to(device) at the reference's benchmark sizes)
HOST_STAGE_MAX = 4096
_TORCH_OF_NP = {np.dtype(np.int32): torch.int32, np.dtype(np.uint32): torch.int32, np.dtype(np.float64): torch.float64,
                np.dtype(np.int64): torch.int64}


class HostOperand:
    """A staged host array standing where a read-only device tensor is expected by the engine's calls.  Valid for the NEXT
    engine call on this thread only (the contract of pai_host_stage): stage, then consume."""
    __slots__ = ("ptr", "shape", "dtype", "device")

    def __init__(self, ptr: int, shape, dtype, device):
        self.ptr, self.shape, self.dtype, self.device = ptr, tuple(shape), dtype, device

    def data_ptr(self) -> int:
        return self.ptr

    def dim(self) -> int:
        return len(self.shape)

    def is_contiguous(self) -> bool:
        return True

    def contiguous(self):
        return self


def host_stage_enabled() -> bool:
    return os.environ.get("PAI_HOST_STAGE", "1") != "0"


def stage_host(arrays, device: torch.device):
    """The host arrays (numpy, C-contiguous; together <= HOST_STAGE_MAX bytes) as HostOperands in one pinned slot."""
    k = len(arrays)
    srcs = (C.c_void_p * k)(*[a.__array_interface__["data"][0] for a in arrays])
    sizes = (C.c_size_t * k)(*[a.nbytes for a in arrays])
    outs = (C.c_void_p * k)()
    rc = _native.load().pai_host_stage(device.index, k, srcs, sizes, _stream(device), outs)
    if rc:
        _native.check(rc)
    return [HostOperand(outs[i] or 0, a.shape, _TORCH_OF_NP[a.dtype], device) for i, a in enumerate(arrays)]


def small_operands(arrays, device: torch.device):
    """Device-readable forms of small host arrays for the next engine call: staged when they fit a slot, uploaded otherwise."""
    arrays = [np.ascontiguousarray(a) for a in arrays]
    if device.index is not None and host_stage_enabled() and \
            sum((a.nbytes + 15) & ~15 for a in arrays) <= HOST_STAGE_MAX and all(a.dtype in _TORCH_OF_NP for a in arrays):
        return stage_host(arrays, device)
    return [torch.from_numpy(a.view(np.int32) if a.dtype == np.uint32 else a).to(device) for a in arrays]


class PublicKeyHandle:
    """Owns a ``pai_pubkey`` on one device."""

    def __init__(self, n: int, key_bits: int, hs: Optional[int], randbits: int, device="cuda:0"):
        self.lib = _native.load()
        self.device = _require_cuda(torch.device(device))
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.n, self.key_bits, self.hs, self.randbits = int(n), int(key_bits), hs, int(randbits)
        self.n_words = (key_bits + 31) // 32
        self.ct_words = 2 * self.n_words
        self.r_words = (randbits + 31) // 32 if hs is not None else self.n_words
        h = C.c_void_p()
        n_w = int_to_words(n, self.n_words)
        if hs is not None:
            hs_w = int_to_words(hs, self.ct_words)
            rc = self.lib.pai_pubkey_create(n_w.ctypes.data_as(C.c_void_p), self.n_words, key_bits,
                                            hs_w.ctypes.data_as(C.c_void_p), self.ct_words, randbits,
                                            self.device.index, C.byref(h))
        else:
            rc = self.lib.pai_pubkey_create(n_w.ctypes.data_as(C.c_void_p), self.n_words, key_bits, None, 0, 0,
                                            self.device.index, C.byref(h))
        _native.check(rc)
        self.h = h

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.pai_pubkey_destroy(self.h)
                self.h = None
        except Exception:
            pass

    # -- helpers ----------------------------------------------------------------------------------
    def empty_ct(self, n: int) -> torch.Tensor:
        return torch.empty((n, self.ct_words), dtype=torch.int32, device=self.device)

    def empty_pt(self, n: int) -> torch.Tensor:
        return torch.empty((n, self.n_words), dtype=torch.int32, device=self.device)

    def _chk(self, t: torch.Tensor, words: int, name: str):
        if t.dtype != torch.int32 or t.dim() != 2 or t.shape[1] != words or not t.is_contiguous() or t.device != self.device:
            raise ValueError(f"{name}: expected contiguous int32 [N,{words}] on {self.device}, got {t.dtype} {tuple(t.shape)} {t.device}")

    # -- hot ops ----------------------------------------------------------------------------------
    def raw_encrypt(self, m: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        self._chk(m, self.n_words, "m")
        out = self.empty_ct(m.shape[0]) if out is None else out
        _native.check(self.lib.pai_raw_encrypt(self.h, _ptr(m), m.shape[0], _ptr(out), _stream(self.device)))
        return out

    def ct_add_plain(self, ct: torch.Tensor, m: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """ct_i * (1 + m_i n) mod n^2 in one pass (pai_ct_add_plain): ciphertext + raw-encrypted plaintext, wire form."""
        self._chk(ct, self.ct_words, "ct")
        self._chk(m, self.n_words, "m")
        if m.shape[0] != ct.shape[0]:
            raise RuntimeError("Size mismatch")
        out = self.empty_ct(ct.shape[0]) if out is None else out
        _native.check(self.lib.pai_ct_add_plain(self.h, _ptr(ct), _ptr(m), ct.shape[0], _ptr(out), _stream(self.device)))
        return out

    def encrypt(self, m: torch.Tensor, r: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        self._chk(m, self.n_words, "m")
        self._chk(r, self.r_words, "r")
        if r.shape[0] != m.shape[0]:
            raise ValueError("m and r must have the same number of rows")
        out = self.empty_ct(m.shape[0]) if out is None else out
        _native.check(self.lib.pai_encrypt(self.h, _ptr(m), _ptr(r), m.shape[0], _ptr(out), _stream(self.device)))
        return out

    def obfuscate_(self, ct: torch.Tensor, r: torch.Tensor) -> torch.Tensor:
        self._chk(ct, self.ct_words, "ct")
        self._chk(r, self.r_words, "r")
        _native.check(self.lib.pai_obfuscate(self.h, _ptr(ct), _ptr(r), ct.shape[0], _stream(self.device)))
        return ct

    def ct_add(self, a: torch.Tensor, b: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        self._chk(a, self.ct_words, "a")
        self._chk(b, self.ct_words, "b")
        bcast = 1 if (b.shape[0] == 1 and a.shape[0] != 1) else 0
        if not bcast and a.shape[0] != b.shape[0]:
            raise RuntimeError("Size mismatch")      # classes.cpp:206 wording
        out = self.empty_ct(a.shape[0]) if out is None else out
        _native.check(self.lib.pai_ct_add(self.h, _ptr(a), _ptr(b), bcast, a.shape[0], _ptr(out), _stream(self.device)))
        return out

    def trim(self) -> int:
        """Frees the fixed-base tables and the grow-only scratch of this handle (rebuilt / re-grown on demand); returns
        the device bytes released.  For processes that hold many keys on one device."""
        v = C.c_size_t(0)
        _native.check(self.lib.pai_pubkey_trim(self.h, C.byref(v)))
        self.__dict__.pop("_dom_consts", None)          # the cached R^k rows go too
        return int(v.value)

    def table_info(self) -> dict:
        """The DJN fixed-base table held right now: {"bytes", "window_bits", "windows"} (zeros before the first obfuscating call)."""
        b, w, j = C.c_size_t(0), C.c_int(0), C.c_int(0)
        _native.check(self.lib.pai_pubkey_table_info(self.h, C.byref(b), C.byref(w), C.byref(j)))
        return {"bytes": int(b.value), "window_bits": int(w.value), "windows": int(j.value)}

    # ---- lazy Montgomery domain (include/paillier_hip.h: pai_ct_mont_mul).  A buffer with tag k holds x R^k mod n^2. ----
    @property
    def mont_bits(self) -> int:
        """log2 of the Montgomery radix R of the ciphertext engine (n^2 geometry)."""
        if getattr(self, "_mont_bits", None) is None:
            v = C.c_int(0)
            _native.check(self.lib.pai_pubkey_mont_bits(self.h, C.byref(v)))
            self._mont_bits = int(v.value)
        return self._mont_bits

    def dom_const(self, k: int) -> torch.Tensor:
        """R^k mod n^2 as one packed device row (k any integer), cached per handle."""
        cache = self.__dict__.setdefault("_dom_consts", {})
        t = cache.pop(k, None)
        if t is None:
            nsq = self.n * self.n
            r = pow(2, self.mont_bits, nsq)
            v = pow(r, k, nsq) if k >= 0 else pow(pow(r, -1, nsq), -k, nsq)
            t = to_device_words(int_to_words(v, self.ct_words)[None, :], self.device)
        cache[k] = t                                   # most recently used last; a small LRU (tags are bounded: paillier.DOM_MAX)
        while len(cache) > 32:
            cache.pop(next(iter(cache)))
        return t

    def ct_mont_mul(self, a: torch.Tensor, b: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """a_i * b_i * R^-1 mod n^2: ONE Montgomery product per element (tags ka, kb -> ka + kb - 1); b of one row broadcasts."""
        self._chk(a, self.ct_words, "a")
        self._chk(b, self.ct_words, "b")
        bcast = 1 if (b.shape[0] == 1 and a.shape[0] != 1) else 0
        if not bcast and a.shape[0] != b.shape[0]:
            raise RuntimeError("Size mismatch")
        out = self.empty_ct(a.shape[0]) if out is None else out
        _native.check(self.lib.pai_ct_mont_mul(self.h, _ptr(a), _ptr(b), bcast, a.shape[0], _ptr(out), _stream(self.device)))
        return out

    def ct_addn(self, ops, raises=None, tag0: int = 0, tag: int = 0, dom_out: int = 0,
                out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """prod_j ops[j][i]^(2^raises[j][i]) mod n^2 in one pass (pai_ct_addn: 2..16 operands, k - 1 products per element).
        ops[0] holds x R^tag0, the others x R^tag, the result x R^dom_out; raises: None, or one int32 [N] device tensor (or
        None) per operand."""
        k = len(ops)
        if not 2 <= k <= 16:
            raise ValueError("ct_addn: between 2 and 16 operands")
        n = ops[0].shape[0]
        for t in ops:
            self._chk(t, self.ct_words, "operand")
            if t.shape[0] != n:
                raise RuntimeError("Size mismatch")
        ptrs = (C.c_void_p * k)(*[t.data_ptr() for t in ops])
        rz = None
        if raises is not None and any(r is not None for r in raises):
            for r in raises:
                if r is not None and (r.dtype != torch.int32 or r.dim() != 1 or r.shape[0] != n or not r.is_contiguous()
                                      or r.device != self.device):
                    raise ValueError("raises: expected contiguous int32 [N] on %s" % self.device)
            rz = (C.c_void_p * k)(*[None if r is None else r.data_ptr() for r in raises])
        out = self.empty_ct(n) if out is None else out
        _native.check(self.lib.pai_ct_addn(self.h, ptrs, rz, k, int(tag0), int(tag), int(dom_out), n, _ptr(out), _stream(self.device)))
        return out

    def ct_retag(self, a: torch.Tensor, k_from: int, k_to: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """x R^k_from -> x R^k_to (one product with the broadcast constant R^(1 + k_to - k_from)); k_to = 0 is the wire form."""
        if k_from == k_to:
            return a
        return self.ct_mont_mul(a, self.dom_const(1 + k_to - k_from), out=out)

    def ct_add_aligned(self, a: torch.Tensor, b: torch.Tensor, delta: torch.Tensor, out: Optional[torch.Tensor] = None,
                       dom: int = 0) -> torch.Tensor:
        """a_i * b_i mod n^2 after raising the lower-exponent side by ^(2^|delta_i|), delta = exponent(a) - exponent(b)
        (int32 [N] on the device): __raw_add with its alignment in one pass (ipcl_python.py:490-526, 570-741).
        dom: the common domain tag of a and b (and of the result)."""
        self._chk(a, self.ct_words, "a")
        self._chk(b, self.ct_words, "b")
        bcast = 1 if (b.shape[0] == 1 and a.shape[0] != 1) else 0
        if not bcast and a.shape[0] != b.shape[0]:
            raise RuntimeError("Size mismatch")
        if isinstance(delta, np.ndarray):
            # host shifts: one call stages them and launches (pai_ct_add_aligned_host) when they fit a slot and the operands are
            # in the wire form; otherwise staged / uploaded here
            delta = np.ascontiguousarray(delta, dtype=np.int32)
            if delta.ndim != 1 or delta.shape[0] != a.shape[0]:
                raise ValueError("delta: expected int32 [N]")
            if dom == 0 and 0 < delta.nbytes <= HOST_STAGE_MAX and host_stage_enabled():
                out = self.empty_ct(a.shape[0]) if out is None else out
                _native.check(self.lib.pai_ct_add_aligned_host(self.h, _ptr(a), _ptr(b), bcast, delta.__array_interface__["data"][0],
                                                               a.shape[0], _ptr(out), _stream(self.device)))
                return out
            if dom != 0:
                self.dom_const(2 - dom)                 # (before the shifts are staged: a staged operand belongs to the very next call)
            delta = small_operands([delta], self.device)[0]
        if delta.dtype != torch.int32 or delta.dim() != 1 or delta.shape[0] != a.shape[0] or not delta.is_contiguous():
            raise ValueError("delta: expected contiguous int32 [N]")
        out = self.empty_ct(a.shape[0]) if out is None else out
        if dom == 0:
            _native.check(self.lib.pai_ct_add_aligned(self.h, _ptr(a), _ptr(b), bcast, _ptr(delta), a.shape[0], _ptr(out),
                                                      _stream(self.device)))
        else:
            _native.check(self.lib.pai_ct_add_aligned_dom(self.h, _ptr(a), _ptr(b), bcast, _ptr(delta), a.shape[0], _ptr(out),
                                                          _ptr(self.dom_const(2 - dom)), _stream(self.device)))
        return out

    def ct_mul(self, ct: torch.Tensor, e: torch.Tensor, ebits_max: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        self._chk(ct, self.ct_words, "ct")
        if isinstance(e, np.ndarray):                   # host exponents: staged and launched in one call when they fit a slot
            e = np.ascontiguousarray(e)
            if e.ndim != 2 or e.dtype not in (np.uint32, np.int32):
                raise ValueError("e: expected uint32 [N or 1, e_words]")
            if 0 < e.nbytes <= HOST_STAGE_MAX and host_stage_enabled():
                bcast = 1 if (e.shape[0] == 1 and ct.shape[0] != 1) else 0
                if not bcast and e.shape[0] != ct.shape[0]:
                    raise RuntimeError("Size mismatch")
                out = self.empty_ct(ct.shape[0]) if out is None else out
                _native.check(self.lib.pai_ct_mul_host(self.h, _ptr(ct), e.__array_interface__["data"][0], e.shape[1], int(ebits_max), bcast,
                                                       ct.shape[0], _ptr(out), _stream(self.device)))
                return out
            e = to_device_words(e, self.device)
        if e.dtype != torch.int32 or e.dim() != 2 or not e.is_contiguous():
            raise ValueError("e: expected contiguous int32 [N or 1, e_words]")
        bcast = 1 if (e.shape[0] == 1 and ct.shape[0] != 1) else 0
        if not bcast and e.shape[0] != ct.shape[0]:
            raise RuntimeError("Size mismatch")
        out = self.empty_ct(ct.shape[0]) if out is None else out
        _native.check(self.lib.pai_ct_mul(self.h, _ptr(ct), _ptr(e), e.shape[1], int(ebits_max), bcast, ct.shape[0],
                                          _ptr(out), _stream(self.device)))
        return out

    def ct_prod(self, ct: torch.Tensor, groups: int = 1, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """[members * groups, W] read member-major -> [groups, W]: out[g] = prod_l ct[l * groups + g] mod n^2
        (the product tree that replaces upstream's pad-rotate-add reduction, ipcl_python.py:810-827)."""
        self._chk(ct, self.ct_words, "ct")
        if groups <= 0 or ct.shape[0] == 0 or ct.shape[0] % groups:
            raise ValueError("ct_prod: the number of rows must be a positive multiple of groups")
        out = self.empty_ct(groups) if out is None else out
        _native.check(self.lib.pai_ct_prod(self.h, _ptr(ct), ct.shape[0], int(groups), _ptr(out), _stream(self.device)))
        return out

    def ct_multiexp(self, ct: torch.Tensor, ct_inv: Optional[torch.Tensor], R: int, K: int, M: int, e: torch.Tensor,
                    ebits_max: int, sign: Optional[torch.Tensor]) -> torch.Tensor:
        """out[r*M + j] = prod_l base(r, l, j)^e[r, l, j] mod n^2 (pai_ct_multiexp); ct [R*K, W], e int32 [R, K, M, ew],
        sign uint8 [K, M] or None.  Raises NativeError(PAI_E_UNSUPPORTED) when the key / sizes are not served."""
        self._chk(ct, self.ct_words, "ct")
        if ct.shape[0] != R * K or e.dtype != torch.int32 or tuple(e.shape[:3]) != (R, K, M) or not e.is_contiguous():
            raise ValueError("ct_multiexp: shape mismatch")
        if (sign is None) != (ct_inv is None):
            raise ValueError("ct_multiexp: signs and inverses come together")
        if sign is not None and (sign.dtype != torch.uint8 or tuple(sign.shape) != (K, M) or not sign.is_contiguous()):
            raise ValueError("ct_multiexp: sign must be contiguous uint8 [K, M]")
        out = self.empty_ct(R * M)
        _native.check(self.lib.pai_ct_multiexp(self.h, _ptr(ct), _ptr(ct_inv), R, K, M, _ptr(e), e.shape[3], int(ebits_max),
                                               _ptr(sign), _ptr(out), _stream(self.device)))
        return out

    def ct_invert(self, ct: torch.Tensor, out: Optional[torch.Tensor] = None, sync: bool = True,
                  flag: Optional[torch.Tensor] = None) -> torch.Tensor:
        """ct^-1 mod n^2.  flag (an int32 device word, see new_flag()): pai_ct_invert_flag — the call returns with the kernels
        queued and a non-invertible input sets bit 0 of THAT word; the API layer keeps it with the container built from the
        result and reads it when the container is exported or decrypted (bindings.ipclCipherText._check).  sync=False without
        a flag: pai_ct_invert_async — the outcome goes to the handle's sticky status word (check_status())."""
        self._chk(ct, self.ct_words, "ct")
        out = self.empty_ct(ct.shape[0]) if out is None else out
        if flag is not None:
            if flag.dtype != torch.int32 or flag.numel() != 1 or flag.device != self.device:
                raise ValueError("flag: expected one int32 word on %s" % self.device)
            _native.check(self.lib.pai_ct_invert_flag(self.h, _ptr(ct), ct.shape[0], _ptr(out), _ptr(flag), _stream(self.device)))
        elif sync:
            _native.check(self.lib.pai_ct_invert(self.h, _ptr(ct), ct.shape[0], _ptr(out), _stream(self.device)))
        else:
            _native.check(self.lib.pai_ct_invert_async(self.h, _ptr(ct), ct.shape[0], _ptr(out), _stream(self.device)))
            self._status_dirty = True
        return out

    def new_flag(self) -> torch.Tensor:
        """A zeroed device word for pai_ct_invert_flag."""
        return torch.zeros(1, dtype=torch.int32, device=self.device)

    def check_status(self, force: bool = False) -> None:
        """Reads (and clears) the sticky status word of this handle's asynchronous calls if one of them ran since the last
        check; raises what the synchronous forms raise.  Synchronises the current stream."""
        if not (force or self.__dict__.get("_status_dirty")):
            return
        self._status_dirty = False
        v = C.c_int(0)
        _native.check(self.lib.pai_pubkey_status(self.h, C.byref(v), 1, _stream(self.device)))
        if v.value & 1:
            raise _native.NativeError(_native.PAI_E_INVALID, "ct_invert: a ciphertext is not invertible modulo n^2")
        if v.value & 2:
            raise _native.NativeError(_native.PAI_E_INVALID, "ct_pow2_hint: max_delta was smaller than a shift of its batch")

    def ct_pow2_(self, ct: torch.Tensor, delta, max_delta: Optional[int] = None) -> torch.Tensor:
        """ct_i <- ct_i^(2^delta_i) in place for delta_i > 0.  delta: int32 device tensor, or a host numpy array (then the
        largest shift is known here and the call never reads anything back: asynchronous for every batch size)."""
        self._chk(ct, self.ct_words, "ct")
        from_host = isinstance(delta, np.ndarray)
        if from_host:
            delta = np.ascontiguousarray(delta, dtype=np.int32).reshape(-1)
            max_delta = int(delta.max()) if delta.size else 0
            delta = small_operands([delta], self.device)[0]
        if delta.dtype != torch.int32 or delta.dim() != 1 or not delta.is_contiguous():
            raise ValueError("delta: expected contiguous int32 [N] or [1]")
        bcast = 1 if (delta.shape[0] == 1 and ct.shape[0] != 1) else 0
        if max_delta is None:
            _native.check(self.lib.pai_ct_pow2(self.h, _ptr(ct), _ptr(delta), bcast, ct.shape[0], _stream(self.device)))
        else:
            _native.check(self.lib.pai_ct_pow2_hint(self.h, _ptr(ct), _ptr(delta), bcast, ct.shape[0], int(max_delta),
                                                    _stream(self.device)))
            if not from_host:
                self._status_dirty = True               # a caller's own hint may be too small: check_status() reports it
        return ct

    # -- data formats either side of the path (device codec, obfuscator randomness) ----------------
    def fp_encode_f64(self, x: torch.Tensor):
        """float64[N] on the device -> (residues int32[N, n_words], exponents int32[N]); fixedpoint.py:54-96.
        The caller has rejected NaN/Inf."""
        if x.dtype != torch.float64 or x.dim() != 1 or not x.is_contiguous() or x.device != self.device:
            raise ValueError("x: expected contiguous float64 [N] on %s" % self.device)
        m = self.empty_pt(x.shape[0])
        expo = torch.empty((x.shape[0],), dtype=torch.int32, device=self.device)
        _native.check(self.lib.pai_fp_encode_f64(self.h, _ptr(x), x.shape[0], _ptr(m), _ptr(expo), _stream(self.device)))
        return m, expo

    def fp_encode_i64(self, x: torch.Tensor):
        """int64[N] on the device -> (residues, exponents = 0); fixedpoint.py:72-74,89-96."""
        if x.dtype != torch.int64 or x.dim() != 1 or not x.is_contiguous() or x.device != self.device:
            raise ValueError("x: expected contiguous int64 [N] on %s" % self.device)
        m = self.empty_pt(x.shape[0])
        expo = torch.empty((x.shape[0],), dtype=torch.int32, device=self.device)
        _native.check(self.lib.pai_fp_encode_i64(self.h, _ptr(x), x.shape[0], _ptr(m), _ptr(expo), _stream(self.device)))
        return m, expo

    def fp_encode_at(self, x: torch.Tensor, target: torch.Tensor):
        """float64[N] or int64[N] -> (residues, exponents) with per-element target exponents (int32[N] or [1]): an element
        whose own exponent is below its target is encoded AT the target (pai_fp_encode_at; the plaintext side of
        ct + plaintext needs no ciphertext squarings then)."""
        if x.dtype not in (torch.float64, torch.int64) or x.dim() != 1 or not x.is_contiguous() or x.device != self.device:
            raise ValueError("x: expected contiguous float64 / int64 [N] on %s" % self.device)
        if target.dtype != torch.int32 or target.dim() != 1 or target.shape[0] not in (1, x.shape[0]) or target.device != self.device:
            raise ValueError("target: expected int32 [N] or [1] on %s" % self.device)
        m = self.empty_pt(x.shape[0])
        expo = torch.empty((x.shape[0],), dtype=torch.int32, device=self.device)
        _native.check(self.lib.pai_fp_encode_at(self.h, _ptr(x), 1 if x.dtype == torch.float64 else 0, x.shape[0], _ptr(target.contiguous()),
                                                1 if target.shape[0] == 1 and x.shape[0] != 1 else 0, _ptr(m), _ptr(expo),
                                                _stream(self.device)))
        return m, expo

    def fp_decode_i64(self, m: torch.Tensor):
        """residues -> (mantissas int64[N], flags int32[N]); flag 1 = element needs the exact host path."""
        self._chk(m, self.n_words, "m")
        mant = torch.empty((m.shape[0],), dtype=torch.int64, device=self.device)
        flag = torch.empty((m.shape[0],), dtype=torch.int32, device=self.device)
        _native.check(self.lib.pai_fp_decode_i64(self.h, _ptr(m), m.shape[0], _ptr(mant), _ptr(flag), _stream(self.device)))
        return mant, flag

    def draw_r(self, n: int, key: bytes, nonce: bytes, counter0: int = 0) -> torch.Tensor:
        """Obfuscator randomness on the device: ChaCha20 key stream under ``key`` (32 bytes, from the OS CSPRNG) and
        ``nonce`` (12 bytes).  DJN keys: r < 2^randbits.  Standard keys: rows of bits(n) random bits (candidates; the
        caller rejects those outside [1, n))."""
        if len(key) != 32 or len(nonce) != 12:
            raise ValueError("ChaCha20 needs a 32-byte key and a 12-byte nonce")
        r = torch.empty((n, self.r_words), dtype=torch.int32, device=self.device)
        # (bytes objects go through ctypes as read-only pointers: the library copies key and nonce before it returns)
        _native.check(self.lib.pai_draw_r(self.h, bytes(key), bytes(nonce), int(counter0) & 0xFFFFFFFF, n, _ptr(r), _stream(self.device)))
        return r

    def random_r(self, n: int, generator: Optional[torch.Generator] = None) -> torch.Tensor:
        """Device-side randomness of the right shape for throughput runs (NOT cryptographic: torch's
        Philox generator).  The Python API draws from the OS CSPRNG instead (paillier.py)."""
        if self.hs is None:
            raise NotImplementedError("random_r is only provided for DJN keys")
        r = torch.randint(-(2**31), 2**31, (n, self.r_words), dtype=torch.int64, device=self.device,
                          generator=generator).to(torch.int32)
        top = self.randbits - 32 * (self.r_words - 1)
        if top < 32:
            r[:, -1] &= (1 << top) - 1
        return r.contiguous()


class PrivateKeyHandle:
    """Owns a ``pai_privkey`` bound to a PublicKeyHandle."""

    def __init__(self, pub: PublicKeyHandle, p: int, q: int):
        self.lib = pub.lib
        self.pub = pub
        self.p, self.q = (p, q) if p < q else (q, p)
        pw = (max(p.bit_length(), q.bit_length()) + 31) // 32
        h = C.c_void_p()
        pa, qa = int_to_words(p, pw), int_to_words(q, pw)
        _native.check(self.lib.pai_privkey_create(pub.h, pa.ctypes.data_as(C.c_void_p), pw,
                                                  qa.ctypes.data_as(C.c_void_p), pw, C.byref(h)))
        self.h = h

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.pai_privkey_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def decrypt(self, ct: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        self.pub._chk(ct, self.pub.ct_words, "ct")
        out = self.pub.empty_pt(ct.shape[0]) if out is None else out
        _native.check(self.lib.pai_decrypt(self.h, _ptr(ct), ct.shape[0], _ptr(out), _stream(self.pub.device)))
        return out


# ------------------------------------------------------------------------------------------------
# handle cache and single-process multi-GPU fan-out
# ------------------------------------------------------------------------------------------------
_pub_cache: "weakref.WeakValueDictionary" = weakref.WeakValueDictionary()
_cache_lock = threading.Lock()


def public_handle(n: int, key_bits: int, hs: Optional[int], randbits: int, device) -> PublicKeyHandle:
    """One PublicKeyHandle per (key material, device) for as long as somebody holds it: every unpickled public key
    or ciphertext of the same key shares the device constants (and, once built, the fixed-base tables)."""
    dev = _require_cuda(torch.device(device))
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    key = (int(n), int(key_bits), None if hs is None else int(hs), int(randbits), dev.index)
    with _cache_lock:
        h = _pub_cache.get(key)
        if h is None:
            h = PublicKeyHandle(n, key_bits, hs, randbits, device=dev)
            _pub_cache[key] = h
        return h


def shard_plan(n: int, nshards: int):
    """[(begin, count)] of the contiguous block partition (pai_shard_plan)."""
    lib = _native.load()
    out = []
    b, c = C.c_size_t(0), C.c_size_t(0)
    for g in range(nshards):
        _native.check(lib.pai_shard_plan(n, nshards, g, C.byref(b), C.byref(c)))
        out.append((int(b.value), int(c.value)))
    return out


def _peer_copy(fn_name: str, shards, hub: torch.Tensor) -> None:
    lib = _native.load()
    k = len(shards)
    devs = (C.c_int32 * k)(*[t.device.index for t in shards])
    ptrs = (C.c_void_p * k)(*[t.data_ptr() for t in shards])
    rows = (C.c_size_t * k)(*[t.shape[0] for t in shards])
    words = hub.shape[1] * hub.element_size() // 4
    for t in shards:
        torch.cuda.synchronize(t.device)          # the copies run on the devices' null streams
    torch.cuda.synchronize(hub.device)
    _native.check(getattr(lib, fn_name)(k, devs, ptrs, rows, words, hub.device.index, C.c_void_p(hub.data_ptr())))


def gather_shards(shards, device) -> torch.Tensor:
    """Row shards on several devices of this process -> one [sum rows, W] tensor on `device` (pai_gather)."""
    total = sum(t.shape[0] for t in shards)
    out = torch.empty((total,) + tuple(shards[0].shape[1:]), dtype=shards[0].dtype, device=device)
    if total:
        _peer_copy("pai_gather", [t.contiguous() for t in shards], out)
    return out


def scatter_shards(src: torch.Tensor, devices) -> list:
    """[N, W] on one device -> contiguous block shards on `devices` (pai_scatter)."""
    plan = shard_plan(src.shape[0], len(devices))
    shards = [torch.empty((c,) + tuple(src.shape[1:]), dtype=src.dtype, device=d) for (_, c), d in zip(plan, devices)]
    if src.shape[0]:
        _peer_copy("pai_scatter", shards, src.contiguous())
    return shards


def fan_out(devices, fn, n_items: int):
    """Runs fn(g, device, begin, count) for every shard of the block partition on its own host thread (ctypes and
    torch release the GIL while the GPU works) and returns the results in shard order; exceptions propagate."""
    plan = shard_plan(n_items, len(devices))
    results = [None] * len(devices)
    errors = []

    def work(g):
        try:
            torch.cuda.set_device(devices[g])
            results[g] = fn(g, devices[g], plan[g][0], plan[g][1])
            torch.cuda.synchronize(devices[g])
        except BaseException as e:      # noqa: BLE001 - re-raised on the caller's thread
            errors.append(e)

    threads = [threading.Thread(target=work, args=(g,)) for g in range(1, len(devices))]
    for t in threads:
        t.start()
    prev = torch.cuda.current_device()
    work(0)
    torch.cuda.set_device(prev)
    for t in threads:
        t.join()
    if errors:
        raise errors[0]
    return results


def profile_enable(on: bool) -> None:
    _native.check(_native.load().pai_profile_enable(1 if on else 0))


def profile_last() -> dict:
    """{kernel name: ms} recorded by the last profiled call on this thread."""
    lib = _native.load()
    out, i = {}, 0
    name = C.create_string_buffer(64)
    ms = C.c_float(0)
    while lib.pai_profile_last(i, name, 64, C.byref(ms)) == 0:
        out[name.value.decode()] = float(ms.value)
        i += 1
    return out
