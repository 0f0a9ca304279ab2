"""Python-visible surface of the reference's native module ``ipcl_python.bindings.ipcl_bindings``
(``bindings/ipcl_bindings.cpp:21-63``), re-created over the HIP engine.

The class and method names are the reference's so that code written against
``ipclPublicKey / ipclPrivateKey / ipclPlainText / ipclCipherText / ipclBigNumber / ipclKeypair /
context / hybridControl / hybridMode`` keeps working, but the containers are device-resident limb
matrices (``engine.py``) instead of ``std::vector<BigNumber>``; a Python ``ipclBigNumber`` object is only
materialised when somebody asks for an individual element (``getTexts()``, ``[i]``).

Wire formats kept bit-for-bit: little-endian bytes padded to a multiple of 4 (``BN2bytes``,
ipcl_bindings.cpp:121-138); public-key tuple ``(scheme, n, bits, hs|0, randbits|0)`` (:66-98);
private-key tuple ``(n, p, q)`` (ipcl_bindings_classes.cpp:142-162); container tuples
``(len, [bytes])`` / ``(len, [bytes], pubkey tuple)`` (:248-265, :356-377); BigNumber ``(bytes,)`` (:481-490).
"""
from __future__ import annotations

import enum
import math
import os
import secrets
import threading
from typing import List, Optional, Sequence, Union

import numpy as np
import torch

# Wire-form additions of at most this many elements return the wire form directly (one launch, pai_ct_add) instead of a lazily
# tagged single product: such batches run on the latency geometry, where a tagged product costs two products anyway, and the
# retag launch at the next boundary (export, decryption, ct * pt) is saved.  0 keeps every addition lazy.
EAGER_ADD_MAX = 1024


def eager_ctct_max(key_bits: int) -> int:
    """Largest ct + ct batch of two wire-form operands that is summed into the wire form at once: the range the library keeps on one
    integer per wavefront (csrc/path_ranges.hpp: lat_add_wire_scale x 4 elements per compute unit; measured on 256 CUs,
    profiles/r06/ctadd_msb_sweep.jsonl) — 2048-bit keys: 25 us at 2 048 and 49 us at 5 120 elements, where a lazily tagged product on
    wave tiles costs ~37 us and its retag as much again."""
    scale = 6 if key_bits <= 1024 else 5 if key_bits <= 2048 else 2 if key_bits <= 3072 else 4 if key_bits <= 4096 else 2
    return scale * EAGER_ADD_MAX

from . import _native, engine

# ------------------------------------------------------------------------------------------------
# BigNumber
# ------------------------------------------------------------------------------------------------


class ipclBigNumber:
    """Value object for one non-negative integer (bindings/ipcl_bindings_classes.cpp:380-491)."""

    __slots__ = ("_v",)

    def __init__(self, data: Union[bytes, bytearray, int, "ipclBigNumber", np.ndarray] = 0):
        if isinstance(data, ipclBigNumber):
            self._v = data._v
        elif isinstance(data, (bytes, bytearray)):
            self._v = int.from_bytes(bytes(data), "little")       # pyByte2BN: little-endian bytes
        elif isinstance(data, np.ndarray):
            self._v = int.from_bytes(np.ascontiguousarray(data, dtype="<u4").tobytes(), "little")
        elif isinstance(data, (int, np.integer)):
            if int(data) < 0:
                raise ValueError("ipclBigNumber: negative values are not supported")
            self._v = int(data)
        elif isinstance(data, (list, tuple)):
            # classes.cpp:386-393: a list of 32-bit words, little-endian
            v = 0
            for i, w in enumerate(data):
                w = int(w)
                if not 0 <= w < 1 << 32:
                    raise TypeError("ipclBigNumber: list elements must be 32-bit unsigned words")
                v |= w << (32 * i)
            self._v = v
        else:
            raise TypeError(f"ipclBigNumber: cannot build from {type(data)}")

    def DwordSize(self) -> int:
        """classes.cpp:458: number of 32-bit words of the value (1 for zero, as BITSIZE_WORD of a one-word BigNumber)."""
        return max(1, (self._v.bit_length() + 31) // 32)

    def BitSize(self) -> int:
        """classes.cpp:459: the bit size of the word array (32 * DwordSize)."""
        return 32 * self.DwordSize()

    def __getitem__(self, n: int) -> int:
        """classes.cpp:422-432: the n-th 32-bit word, little-endian; IndexError (std::out_of_range) past the end."""
        n = int(n)
        length = self.DwordSize()
        if not 0 <= n < length:
            raise IndexError("Index is larger than size: %d" % length)
        return (self._v >> (32 * n)) & 0xFFFFFFFF

    def data(self):
        """classes.cpp:460-471: (word count, [words])."""
        length = self.DwordSize()
        return (length, [(self._v >> (32 * i)) & 0xFFFFFFFF for i in range(length)])

    def to_bytes(self) -> bytes:
        """BN2bytes: little-endian, length = ceil(bits/32)*4 (0 encodes as 4 zero bytes upstream)."""
        words = max(1, (self._v.bit_length() + 31) // 32)
        return self._v.to_bytes(4 * words, "little")

    def __int__(self):
        return self._v

    __index__ = __int__

    def __eq__(self, other):
        if isinstance(other, ipclBigNumber):
            return self._v == other._v
        if isinstance(other, (int, np.integer)):
            return self._v == int(other)
        return NotImplemented

    def __hash__(self):
        return hash(self._v)

    def __repr__(self):
        return str(self._v)

    __str__ = __repr__

    def __getstate__(self):
        return (self.to_bytes(),)

    def __setstate__(self, state):
        self._v = int.from_bytes(state[0], "little")

    # the arithmetic operators of classes.cpp:433-457 (not used by the L3 API)
    def __add__(self, o):
        return ipclBigNumber(self._v + int(o))

    def __sub__(self, o):
        return ipclBigNumber(self._v - int(o))

    def __mul__(self, o):
        return ipclBigNumber(self._v * int(o))

    def __mod__(self, o):
        return ipclBigNumber(self._v % int(o))

    def __lt__(self, o):
        return self._v < int(o)

    def __le__(self, o):
        return self._v <= int(o)

    def __gt__(self, o):
        return self._v > int(o)

    def __ge__(self, o):
        return self._v >= int(o)


ipclBigNumber.Zero = ipclBigNumber(0)
ipclBigNumber.One = ipclBigNumber(1)
ipclBigNumber.Two = ipclBigNumber(2)


def _as_int(x) -> int:
    return int(x)


# ------------------------------------------------------------------------------------------------
# keys
# ------------------------------------------------------------------------------------------------


def _default_device() -> torch.device:
    if not torch.cuda.is_available():
        raise RuntimeError("pailliercryptolib_python_amd needs an AMD GPU (gfx950); there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def _env_devices() -> Optional[List[torch.device]]:
    """PAI_DEVICES=all | "0,1,2": the devices new keys fan their heavy element-wise operations out to (single-process
    multi-GPU; one-process-per-GPU jobs use sharding.py instead).  Unset: the current device only."""
    spec = os.environ.get("PAI_DEVICES", "").strip()
    if not spec:
        return None
    if spec == "all":
        return [torch.device("cuda", i) for i in range(torch.cuda.device_count())]
    return [torch.device("cuda", int(tok)) for tok in spec.split(",") if tok.strip() != ""]


# below this many elements per device a batch stays on the home device (launch + peer-copy latency dominates)
FANOUT_MIN_PER_DEVICE = int(os.environ.get("PAI_FANOUT_MIN", "2048"))


def _random_unit(n: int) -> int:
    while True:
        x = secrets.randbelow(n - 2) + 2
        if math.gcd(x, n) == 1:
            return x


def _rows_not_in_1_n(r: torch.Tensor, n_words: torch.Tensor) -> torch.Tensor:
    """[N] bool: row i of the limb matrix r (int32 bit patterns of uint32 limbs, little-endian) is 0 or >= n."""
    u = r.to(torch.int64) & 0xFFFFFFFF
    lt = torch.zeros(r.shape[0], dtype=torch.bool, device=r.device)
    eq = torch.ones(r.shape[0], dtype=torch.bool, device=r.device)
    for w in range(r.shape[1] - 1, -1, -1):
        lt |= eq & (u[:, w] < n_words[w])
        eq &= u[:, w] == n_words[w]
    return ~lt | (u.sum(dim=1) == 0)


class ipclPublicKey:
    """Replaces the pybind class at bindings/ipcl_bindings_classes.cpp:12-91."""

    def __init__(self, n: Union[ipclBigNumber, int], bits: int = 1024, enableDJN: bool = False, *,
                 hs: Optional[int] = None, randbits: Optional[int] = None, device=None, devices=None):
        self._n = _as_int(n)
        self._bits = int(bits)
        if self._n.bit_length() > self._bits:
            raise RuntimeError("ipclPublicKey: n is wider than the declared key length")
        self._djn = bool(enableDJN) or hs is not None
        if self._djn:
            if hs is None:
                # upstream DJN set-up: hs = (-x^2)^n mod n^2 for a random unit x
                nsq = self._n * self._n
                x = _random_unit(self._n)
                # pai_host_modexp serves odd moduli up to 260 words (n^2 of keys up to 4160 bits); beyond: CPython's pow
                native_ok = self._n & 1 and nsq.bit_length() <= 32 * _native.HOST_MODEXP_MAX_WORDS
                hs = _native.host_modexp((-x * x) % nsq, self._n, nsq) if native_ok else pow((-x * x) % nsq, self._n, nsq)
            self._hs = int(hs)
            self._randbits = int(randbits) if randbits is not None else self._bits // 2
        else:
            self._hs, self._randbits = None, 0
        self._devices: Optional[List[torch.device]] = None
        if devices is not None:
            self.set_devices(devices)
        elif device is not None:
            self._devices = [torch.device(device)]
        self._handles: dict = {}
        self._obf_pool: Optional[torch.Tensor] = None
        self._obf_lock = threading.Lock()        # pooled obfuscators are single-use: take and fill are atomic

    # -- lazily created device handles (cached per key material and device: engine.public_handle) ----
    def _device_list(self) -> List[torch.device]:
        if self._devices is None:
            self._devices = _env_devices() or [_default_device()]
        if any(d.index is None for d in self._devices):
            self._devices = [d if d.index is not None else _default_device() for d in self._devices]
        return self._devices

    def set_devices(self, devices) -> None:
        """Extension: the devices of this process the key lives on.  The first one is the home device (containers
        are resident there); encrypt / decrypt / ct*pt of large batches are sharded over all of them by contiguous
        blocks and gathered back with peer copies."""
        devs = [torch.device(d) for d in devices]
        devs = [torch.device("cuda", d.index if d.index is not None else torch.cuda.current_device()) for d in devs]
        if not devs:
            raise ValueError("set_devices: need at least one device")
        self._devices = devs
        self.__dict__.pop("_home_handle", None)

    def handle_on(self, device: torch.device) -> engine.PublicKeyHandle:
        h = self._handles.get(device.index)
        if h is None:
            h = engine.public_handle(self._n, self._bits, self._hs, self._randbits, device)
            self._handles[device.index] = h
        return h

    @property
    def handle(self) -> engine.PublicKeyHandle:
        """The handle on the home device."""
        h = self.__dict__.get("_home_handle")
        if h is None:
            h = self.handle_on(self._device_list()[0])
            self.__dict__["_home_handle"] = h
        return h

    @property
    def device(self) -> torch.device:
        return self.handle.device

    def trim(self) -> int:
        """Extension (many keys per device): releases the fixed-base tables and grow-only scratch of every device handle of
        this key (pai_pubkey_trim) — they are rebuilt / re-grown on demand.  Returns the device bytes released."""
        return sum(h.trim() for h in self._handles.values())

    def fanout_devices(self, n_items: int) -> Optional[List[torch.device]]:
        """The device list to shard a batch of n_items over, or None when it should stay on the home device."""
        devs = self._device_list()
        if len(devs) < 2 or n_items < FANOUT_MIN_PER_DEVICE * len(devs):
            return None
        return devs

    # -- reference surface -------------------------------------------------------------------------
    @property
    def n(self) -> ipclBigNumber:
        return ipclBigNumber(self._n)

    @property
    def length(self) -> int:
        return self._bits

    @property
    def nsquare(self) -> ipclBigNumber:
        return ipclBigNumber(self._n * self._n)

    def isDJN(self) -> bool:
        return self._djn

    def __repr__(self):
        return "<ipclPublicKey %s>" % str(hash(self))[:10]

    def __eq__(self, other):
        # upstream compares the underlying object identity (App. A); value equality is the evident intent
        return isinstance(other, ipclPublicKey) and self._n == other._n

    def __hash__(self):
        return hash(("ipclPublicKey", self._n))

    def __getstate__(self):
        if self._djn:
            return (1, ipclBigNumber(self._n).to_bytes(), self._bits, ipclBigNumber(self._hs).to_bytes(), self._randbits)
        return (0, ipclBigNumber(self._n).to_bytes(), self._bits, 0, 0)

    def __setstate__(self, t):
        scheme, n_b, bits = t[0], t[1], t[2]
        self._n = int.from_bytes(n_b, "little")
        self._bits = int(bits)
        self._djn = scheme != 0
        if self._djn:
            self._hs = int.from_bytes(t[3], "little")
            self._randbits = int(t[4])
        else:
            self._hs, self._randbits = None, 0
        self._devices = None
        self._handles = {}
        self._obf_pool = None
        self._obf_lock = threading.Lock()

    # randomness for the obfuscator: OS CSPRNG on the host, expanded / uploaded as limbs
    def _draw_r(self, count: int, h: Optional[engine.PublicKeyHandle] = None) -> torch.Tensor:
        h = h or self.handle
        if self._djn:
            # fresh 256-bit key + 96-bit nonce from the OS CSPRNG, expanded on the device (ChaCha20, RFC 8439)
            return h.draw_r(count, secrets.token_bytes(32), secrets.token_bytes(12))
        # standard scheme: r uniform in [1, n) by rejection sampling on the device — candidates of bits(n) random bits
        # from the ChaCha20 stream (a fresh OS-CSPRNG key per round), a limb-wise comparison with n, redraw of the
        # rejected rows (each round keeps more than half of them)
        n_w = torch.from_numpy(engine.int_to_words(self._n, h.n_words).astype(np.int64)).to(h.device)
        r = h.draw_r(count, secrets.token_bytes(32), secrets.token_bytes(12))
        for _ in range(128):
            bad = _rows_not_in_1_n(r, n_w)
            nbad = int(bad.sum())
            if nbad == 0:
                return r
            idx = torch.nonzero(bad, as_tuple=False).reshape(-1)
            r[idx] = h.draw_r(nbad, secrets.token_bytes(32), secrets.token_bytes(12))
        raise RuntimeError("standard-scheme randomness: rejection sampling did not terminate")

    # -- obfuscator pool: obfuscators hs^r (or r^n) computed ahead of time, each used once ---------
    def fill_obfuscator_pool(self, count: int) -> None:
        """Extension: precomputes `count` fresh obfuscators on the home device (randomness from the same CSPRNG-keyed
        source as encrypt).  Later encryptions take theirs from the pool — an encryption is then the raw form times
        one pooled obfuscator, i.e. ONE modular multiplication instead of the fixed-base exponentiation (the
        reference's `apply_obfuscator=False` fast path followed by re-obfuscation, ipcl_python.py:103-106,342-346, done
        ahead of time).  Every pooled value is consumed exactly once; a batch larger than the pool encrypts directly."""
        h = self.handle
        one = torch.zeros((count, h.n_words), dtype=torch.int32, device=h.device)      # E(0; r) = 1 * obf(r)
        fresh = h.encrypt(one, self._draw_r(count))
        with self._obf_lock:
            pool = self._obf_pool
            self._obf_pool = fresh if pool is None or pool.shape[0] == 0 else torch.cat([pool, fresh], dim=0)

    def obfuscator_pool_size(self) -> int:
        with self._obf_lock:
            return 0 if self._obf_pool is None else int(self._obf_pool.shape[0])

    def _take_obfuscators(self, count: int) -> Optional[torch.Tensor]:
        """`count` pooled obfuscators, removed from the pool under the key's lock: two concurrent encryptions can never
        receive the same value (reuse would make ct1 * ct2^-1 = 1 + (m1 - m2) n, i.e. leak the plaintext difference)."""
        if count == 0:
            return None
        with self._obf_lock:
            pool = self._obf_pool
            if pool is None or pool.shape[0] < count:
                return None
            taken, self._obf_pool = pool[:count], pool[count:]
        return taken.contiguous()

    def encrypt_words(self, m: torch.Tensor, make_secure: bool = True, r: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Residues [N, n_words] on the home device -> ciphertexts [N, ct_words] on the home device; large batches
        are sharded over the key's devices (scatter, encrypt per device on its own thread, gather)."""
        h = self.handle
        if make_secure and r is None:
            obf = self._take_obfuscators(m.shape[0])
            if obf is not None:
                return h.ct_add(h.raw_encrypt(m), obf)             # (1 + m n) * obf mod n^2: the same bits as a direct encryption
        devs = self.fanout_devices(m.shape[0])
        if devs is None:
            if not make_secure:
                return h.raw_encrypt(m)
            return h.encrypt(m, self._draw_r(m.shape[0]) if r is None else r)
        m_sh = engine.scatter_shards(m, devs)
        r_sh = engine.scatter_shards(r, devs) if (make_secure and r is not None) else None

        def work(g, dev, begin, count):
            hg = self.handle_on(dev)
            if count == 0:
                return hg.empty_ct(0)
            if not make_secure:
                return hg.raw_encrypt(m_sh[g])
            return hg.encrypt(m_sh[g], self._draw_r(count, hg) if r_sh is None else r_sh[g])

        return engine.gather_shards(engine.fan_out(devs, work, m.shape[0]), h.device)

    def ct_mul_words(self, ct: torch.Tensor, e: torch.Tensor, ebits_max: int) -> torch.Tensor:
        """ct_i ^ e_i mod n^2 (classes.cpp:324-325) on the home device's limb matrices, sharded when large."""
        h = self.handle
        devs = self.fanout_devices(ct.shape[0])
        if isinstance(e, np.ndarray) and devs is not None:
            e = engine.to_device_words(e, h.device)
        if devs is None:          # (host exponents of a small batch reach the kernel through a pinned slot: engine.ct_mul)
            return h.ct_mul(ct, e, ebits_max)
        ct_sh = engine.scatter_shards(ct, devs)
        bcast = e.shape[0] == 1 and ct.shape[0] != 1
        e_sh = [e.to(d) for d in devs] if bcast else engine.scatter_shards(e, devs)

        def work(g, dev, begin, count):
            hg = self.handle_on(dev)
            return hg.ct_mul(ct_sh[g], e_sh[g], ebits_max) if count else hg.empty_ct(0)

        return engine.gather_shards(engine.fan_out(devs, work, ct.shape[0]), h.device)

    def encrypt(self, pt: "ipclPlainText", make_secure: bool = True, *, r: Optional[torch.Tensor] = None) -> "ipclCipherText":
        """classes.cpp:53-60.  ``r`` (extension) injects the obfuscator randomness for reproducible runs."""
        return ipclCipherText(self, self.encrypt_words(pt._device_words(self.handle), make_secure, r))

    def encrypt_tolist(self, pt: "ipclPlainText", make_secure: bool = True, *, r: Optional[torch.Tensor] = None) -> list:
        """classes.cpp:61-70: encrypt, then the ciphertexts as a list of ipclBigNumber (CipherText.getTexts())."""
        return self.encrypt(pt, make_secure, r=r).getTexts()

    def apply_obfuscator(self, x, *, r: Optional[torch.Tensor] = None):
        """classes.cpp:71-83: BigNumber -> BigNumber, CipherText -> list of BigNumber (as upstream)."""
        h = self.handle
        if isinstance(x, (ipclBigNumber, int)):
            ct = engine.to_device_words(engine.ints_to_words([_as_int(x)], h.ct_words), h.device)
            h.obfuscate_(ct, self._draw_r(1) if r is None else r)
            return ipclBigNumber(engine.to_host_words(ct)[0])
        if isinstance(x, ipclCipherText):
            ct = x._t.clone()
            h.obfuscate_(ct, self._draw_r(ct.shape[0]) if r is None else r)
            return ipclCipherText(self, ct, taint=x._taint).getTexts()
        raise TypeError("apply_obfuscator: expected ipclBigNumber or ipclCipherText")


class ipclPrivateKey:
    """Replaces the pybind class at bindings/ipcl_bindings_classes.cpp:93-163."""

    def __init__(self, pk: Union[ipclPublicKey, ipclBigNumber, int], p, q, _q=None):
        if isinstance(pk, ipclPublicKey):
            self._pk = pk
        else:
            # (n, p, q) form used by the pickle constructor (classes.cpp:161)
            n = _as_int(pk)
            self._pk = ipclPublicKey(n, n.bit_length(), False)
        p, q = _as_int(p), _as_int(q)
        self._p, self._q = (p, q) if p < q else (q, p)
        if self._p * self._q != self._pk._n:
            raise RuntimeError("ipclPrivateKey: p*q does not match the public key")
        self._handles: dict = {}

    def handle_on(self, device: torch.device) -> engine.PrivateKeyHandle:
        h = self._handles.get(device.index)
        if h is None:
            h = engine.PrivateKeyHandle(self._pk.handle_on(device), self._p, self._q)
            self._handles[device.index] = h
        return h

    @property
    def handle(self) -> engine.PrivateKeyHandle:
        return self.handle_on(self._pk._device_list()[0])

    def decrypt_words(self, ct: torch.Tensor) -> torch.Tensor:
        """Ciphertexts [N, ct_words] -> residues [N, n_words] on the home device, sharded over the key's devices
        when the batch is large."""
        hpub = self.handle.pub
        words = ct if ct.device == hpub.device else ct.to(hpub.device)
        devs = self._pk.fanout_devices(words.shape[0])
        if devs is None:
            return self.handle.decrypt(words)
        ct_sh = engine.scatter_shards(words, devs)

        def work(g, dev, begin, count):
            hg = self.handle_on(dev)
            return hg.decrypt(ct_sh[g]) if count else hg.pub.empty_pt(0)

        return engine.gather_shards(engine.fan_out(devs, work, words.shape[0]), hpub.device)

    @property
    def n(self):
        return ipclBigNumber(self._pk._n)

    @property
    def p(self):
        return ipclBigNumber(self._p)

    @property
    def q(self):
        return ipclBigNumber(self._q)

    def __repr__(self):
        return "<ipclPrivateKey %s>" % str(hash(self))[:10]

    def __hash__(self):
        return hash(("ipclPrivateKey", self._p, self._q))

    def __eq__(self, other):
        return isinstance(other, ipclPrivateKey) and (self._p, self._q) == (other._p, other._q)

    def decrypt(self, ct: "ipclCipherText") -> "ipclPlainText":
        """classes.cpp:127-133."""
        if ct.public_key._n != self._pk._n:
            raise RuntimeError("ipclPrivateKey.decrypt: public key mismatch")
        ct._check()                                          # a failed asynchronous inversion is raised here, on its own result
        return ipclPlainText(self.decrypt_words(ct._t))

    def decrypt_tolist(self, ct: "ipclCipherText") -> list:
        """classes.cpp:134-141: decrypt, then the residues as a list of ipclBigNumber (PlainText.getTexts())."""
        return self.decrypt(ct).getTexts()

    def __getstate__(self):
        return (ipclBigNumber(self._pk._n).to_bytes(), ipclBigNumber(self._p).to_bytes(), ipclBigNumber(self._q).to_bytes())

    def __setstate__(self, t):
        n, p, q = (int.from_bytes(b, "little") for b in t)
        self._pk = ipclPublicKey(n, n.bit_length(), False)
        self._p, self._q = (p, q) if p < q else (q, p)
        self._handles = {}


# ------------------------------------------------------------------------------------------------
# containers
# ------------------------------------------------------------------------------------------------


def _slice_bounds(key: slice, length: int):
    start, stop, step = key.indices(length)
    if step != 1:
        raise RuntimeError("Step size not supported")        # classes.cpp:223,315
    return start, max(stop, start)


class _Container:
    """Common part of ipclPlainText / ipclCipherText: a [N, words] int32 tensor (uint32 bit patterns)
    on a device, or — before a key is known — a host list of ints."""

    def __len__(self):
        return self.getSize()

    def getSize(self) -> int:
        return int(self._t.shape[0]) if self._t is not None else len(self._ints)

    def getTexts(self) -> List[ipclBigNumber]:
        if self._t is None:
            return [ipclBigNumber(v) for v in self._ints]
        return [ipclBigNumber(v) for v in engine.words_to_ints(engine.to_host_words(self._t))]

    def getElementVec(self, i: int) -> List[int]:
        return engine.to_host_words(self._row(i))[0].tolist()

    def getElementHex(self, i: int) -> str:
        return "%X" % int(self[i])

    def _row(self, i: int):
        n = self.getSize()
        if not 0 <= i < n:
            raise IndexError("index out of range")
        return self._t[i:i + 1]


class ipclPlainText(_Container):
    """bindings/ipcl_bindings_classes.cpp:165-266."""

    def __init__(self, data=None):
        self._t: Optional[torch.Tensor] = None
        self._ints: List[int] = []
        if data is None:
            return
        if isinstance(data, torch.Tensor):
            self._t = data
        elif isinstance(data, (ipclBigNumber, int, np.integer)):
            self._ints = [_as_int(data)]
        elif isinstance(data, np.ndarray) and data.dtype == np.uint32 and data.ndim == 2:
            self._ints = engine.words_to_ints(data)
        else:
            self._ints = [_as_int(v) for v in data]

    def _device_words(self, h: engine.PublicKeyHandle) -> torch.Tensor:
        if self._t is not None:
            if self._t.shape[1] != h.n_words:
                raise RuntimeError("ipclPlainText: width does not match the key")
            return self._t if self._t.device == h.device else self._t.to(h.device)
        for v in self._ints:
            if v.bit_length() > 32 * h.n_words:
                raise RuntimeError("ipclPlainText: value wider than the key")
        return engine.to_device_words(engine.ints_to_words(self._ints, h.n_words), h.device)

    def getTexts(self):
        return super().getTexts()

    def __getitem__(self, key):
        if isinstance(key, slice):
            a, b = _slice_bounds(key, len(self))
            return ipclPlainText(engine.rows_slice(self._t, a, b - a)) if self._t is not None else ipclPlainText(self._ints[a:b])
        if self._t is None:
            return ipclBigNumber(self._ints[key])
        return ipclBigNumber(engine.to_host_words(self._row(int(key)))[0])

    def rotate(self, shift: int) -> "ipclPlainText":
        n = len(self)
        k = shift % n if n else 0
        if self._t is None:
            return ipclPlainText(self._ints[k:] + self._ints[:k])
        return ipclPlainText(engine.rows_rotate(self._t, k))

    def __eq__(self, other):
        if len(self) != len(other):
            raise RuntimeError("Size mismatch")
        if [int(a) for a in self.getTexts()] != [int(b) for b in other.getTexts()]:
            raise RuntimeError("PlainText mismatch")
        return True

    def __repr__(self):
        return "<ipclPlainText %s>" % str(id(self))[:10]

    def __getstate__(self):
        return (len(self), [b.to_bytes() for b in self.getTexts()])

    def __setstate__(self, t):
        self._t = None
        self._ints = [int.from_bytes(b, "little") for b in t[1]]


def merge_taint(*taints) -> tuple:
    """Union (by identity) of the outcome words of several containers (ipclCipherText._taint)."""
    out, seen = [], set()
    for t in taints:
        for f in t:
            if id(f) not in seen:
                seen.add(id(f))
                out.append(f)
    return tuple(out)


class ipclCipherText(_Container):
    """bindings/ipcl_bindings_classes.cpp:268-378: ciphertext container bound to a public key."""

    def __init__(self, pubkey: ipclPublicKey, data=None, *, dom: int = 0, taint: tuple = ()):
        self._pk = pubkey
        self._ints: List[int] = []
        # Outcome words of the asynchronous inversions this container's rows were computed from (pai_ct_invert_flag: one int32
        # device word per call, bit 0 = an input was not invertible).  They travel with every container derived from this one
        # and are read — one synchronisation, then dropped — when rows leave the device: _check().
        self._taint: tuple = tuple(taint)
        self._dev: Optional[torch.Tensor] = None      # limb matrix on the key's home device ...
        self._host: Optional[np.ndarray] = None       # ... or host words that have not been needed on a device yet
        # Lazy Montgomery domain (extension): the device rows hold x R^_dom mod n^2.  Additions are ONE
        # Montgomery product (tags ka, kb -> ka + kb - 1); whoever needs the wire form (getTexts, pickling, decryption,
        # ct * pt, every `_t` / `words` access) gets it through one more product, done once and cached in place.
        self._dom = 0
        self._lock = threading.Lock()                 # (_dev, _dom) change together: upload, device move, retag
        W = 2 * ((pubkey._bits + 31) // 32)
        if isinstance(data, torch.Tensor):
            if data.shape[1] != W:
                raise RuntimeError("ipclCipherText: width does not match the key")
            self._dev = data
            self._dom = int(dom)
        elif isinstance(data, ipclCipherText):
            self._dev, self._host, self._dom = data._dev, data._host, data._dom
            self._taint = merge_taint(self._taint, data._taint)
        elif isinstance(data, np.ndarray) and data.dtype == np.uint32 and data.ndim == 2:
            if data.shape[1] != W:
                raise RuntimeError("ipclCipherText: width does not match the key")
            self._host = np.ascontiguousarray(data)
        else:
            if isinstance(data, (ipclBigNumber, int, np.integer)):
                vals = [_as_int(data)]
            else:
                vals = [_as_int(v) for v in (data if data is not None else [])]
            for v in vals:
                if v.bit_length() > 32 * W:
                    raise RuntimeError("ipclCipherText: width does not match the key")
            self._host = engine.ints_to_words(vals, W) if vals else np.zeros((0, W), dtype=np.uint32)

    def _raw(self):
        """(limb matrix on the home device, domain tag): rows hold x R^tag mod n^2.  Host-built containers (pickles, lists
        of BigNumbers) are uploaded — and the key's device handle created — only when an operation first needs them."""
        with self._lock:
            if self._dev is None:
                self._dev = engine.to_device_words(self._host, self._pk.handle.device)
                self._host = None
            elif self._dev.device != self._pk.handle.device:
                self._dev = self._dev.to(self._pk.handle.device)
            return self._dev, self._dom

    @property
    def _t(self) -> torch.Tensor:
        """The limb matrix on the home device in the wire form (canonical residues of the ciphertexts themselves).  The
        retag (one product, cached in place) happens under the container's lock: a second thread can never pair the
        retagged rows with the old tag."""
        self._raw()
        with self._lock:
            if self._dom != 0:
                self._dev = self._pk.handle.ct_retag(self._dev, self._dom, 0)
                self._dom = 0
            return self._dev

    def _check(self) -> None:
        """Raises if an asynchronous inversion behind these rows met a non-invertible ciphertext (the rows are then undefined):
        nothing computed by a failed call leaves the device, and the error is raised on the object it belongs to, every time
        that object is exported.  A clean outcome is final, so the words are dropped after the first look."""
        if self._taint:
            for f in self._taint:
                if int(f.item()) & 1:
                    raise _native.NativeError(_native.PAI_E_INVALID, "ct_invert: a ciphertext this result was computed from is "
                                                                     "not invertible modulo n^2")
            self._taint = ()

    def getSize(self) -> int:
        return int(self._host.shape[0]) if self._dev is None else int(self._dev.shape[0])

    def getTexts(self) -> List[ipclBigNumber]:
        self._check()
        words = self._host if self._dev is None else engine.to_host_words(self._t)
        return [ipclBigNumber(v) for v in engine.words_to_ints(words)]

    @property
    def public_key(self) -> ipclPublicKey:
        return self._pk

    @property
    def words(self) -> torch.Tensor:
        """Device tensor [N, ct_words] int32 (extension: direct access to the limb matrix; the outcome of pending
        asynchronous inversions is checked first — internal callers that stay on the device use `_t`)."""
        self._check()
        return self._t

    def getCipherText(self):
        return self

    def __getitem__(self, key):
        if isinstance(key, slice):
            a, b = _slice_bounds(key, len(self))
            t, dom = self._raw()
            return ipclCipherText(self._pk, engine.rows_slice(t, a, b - a), dom=dom, taint=self._taint)
        self._check()
        return ipclBigNumber(engine.to_host_words(self._row(int(key)))[0])

    def getElementVec(self, i: int) -> List[int]:
        self._check()
        return super().getElementVec(i)

    def rotate(self, shift: int) -> "ipclCipherText":
        n = len(self)
        k = shift % n if n else 0
        t, dom = self._raw()
        return ipclCipherText(self._pk, engine.rows_rotate(t, k), dom=dom, taint=self._taint)

    def __add__(self, other):
        h = self._pk.handle
        if isinstance(other, ipclPlainText):
            other = self._pk.encrypt(other, False)
        if not isinstance(other, ipclCipherText):
            return NotImplemented
        if len(other) != len(self) and len(other) != 1:
            raise RuntimeError("Size mismatch")
        (ta, ka), (tb, kb) = self._raw(), other._raw()
        if ka == 0 and kb == 0 and ta.shape[0] <= eager_ctct_max(self._pk._bits):
            # small wire-form operands: the wire form at once (the latency geometry spends two products on a tagged result too)
            return ipclCipherText(self._pk, h.ct_add(ta, tb), taint=merge_taint(self._taint, other._taint))
        return ipclCipherText(self._pk, h.ct_mont_mul(ta, tb), dom=ka + kb - 1,        # one product; the tag remembers the R^-1
                              taint=merge_taint(self._taint, other._taint))

    def __mul__(self, other: ipclPlainText):
        if not isinstance(other, ipclPlainText):
            return NotImplemented
        h = self._pk.handle
        vals = [int(v) for v in other.getTexts()]
        if len(vals) != len(self) and len(vals) != 1:
            raise RuntimeError("Size mismatch")
        bits = max(1, max(v.bit_length() for v in vals))
        ew = (bits + 31) // 32
        e = engine.to_device_words(engine.ints_to_words(vals, ew), h.device)
        return ipclCipherText(self._pk, self._pk.ct_mul_words(self._t, e, bits), taint=self._taint)

    def __repr__(self):
        return "<ipclCipherText %s>" % str(id(self))[:10]

    __str__ = __repr__

    def __getstate__(self):
        return (len(self), [b.to_bytes() for b in self.getTexts()], self._pk.__getstate__())

    def __setstate__(self, t):
        pk = ipclPublicKey.__new__(ipclPublicKey)
        pk.__setstate__(t[2])
        self.__init__(pk, [int.from_bytes(b, "little") for b in t[1]])


# ------------------------------------------------------------------------------------------------
# key generation (host) and the QAT/hybrid shims
# ------------------------------------------------------------------------------------------------


def _is_probable_prime(n: int, rounds: int = 32) -> bool:
    if n < 2:
        return False
    for s in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37, 41, 43, 47, 53, 59, 61, 67, 71, 73, 79, 83, 89, 97):
        if n % s == 0:
            return n == s
    d, r = n - 1, 0
    while d % 2 == 0:
        d //= 2
        r += 1
    for _ in range(rounds):
        a = secrets.randbelow(n - 3) + 2
        x = pow(a, d, n)
        if x in (1, n - 1):
            continue
        for _ in range(r - 1):
            x = x * x % n
            if x == n - 1:
                break
        else:
            return False
    return True


def _random_prime(bits: int, congruent_3_mod_4: bool) -> int:
    while True:
        c = secrets.randbits(bits) | (1 << (bits - 1)) | (1 << (bits - 2)) | 1
        if congruent_3_mod_4:
            c |= 3
        if _is_probable_prime(c):
            return c


def _djn_hs_from_primes(p: int, q: int) -> int:
    """hs = (-x^2)^n mod n^2 for a random unit x (the DJN set-up of ipclPublicKey.__init__) with the primes in hand: the two
    powers modulo p^2 and q^2 (exponent reduced modulo the group orders, a quarter of the limb products each) on two host
    threads, then the CRT — the same integer as the single power modulo n^2."""
    n = p * q
    x = _random_unit(n)
    base = (-x * x) % (n * n)
    mods = (p * p, q * q)
    exps = (n % (p * (p - 1)), n % (q * (q - 1)))
    native_ok = all(m.bit_length() <= 32 * _native.HOST_MODEXP_MAX_WORDS for m in mods)
    out = [0, 0]

    def work(i: int) -> None:
        out[i] = _native.host_modexp(base % mods[i], exps[i], mods[i]) if native_ok else pow(base % mods[i], exps[i], mods[i])

    th = threading.Thread(target=work, args=(1,))
    th.start()
    work(0)
    th.join()
    a, b = out
    return a + mods[0] * (((b - a) * pow(mods[0], -1, mods[1])) % mods[1])


class ipclKeypair:
    @staticmethod
    def generate_keypair(n_length: int = 1024, enable_DJN: bool = True):
        """ipcl::generateKeypair (bindings/ipcl_bindings.cpp:12-15).  DJN keys use p = q = 3 (mod 4) with
        gcd(p-1, q-1) = 2 (upstream's constraint)."""
        n_length = int(n_length)
        if n_length < 64 or n_length % 4 != 0:
            raise RuntimeError("generate_keypair: n_length must be a multiple of 4 and at least 64")
        half = n_length // 2
        while True:
            if n_length % 64 == 0 and 128 <= n_length <= _native.KEYGEN_MAX_BITS:
                # native search (pai_keygen): sieve + Miller-Rabin on the host cores, both primes in parallel
                p, q = _native.keygen(n_length, enable_DJN)
            else:
                p = _random_prime(half, enable_DJN)
                q = _random_prime(half, enable_DJN)
            if p == q or (p * q).bit_length() != n_length:
                continue
            if enable_DJN and math.gcd(p - 1, q - 1) != 2:
                continue
            if math.gcd(p * q, (p - 1) * (q - 1)) != 1:
                continue
            break
        hs = _djn_hs_from_primes(p, q) if enable_DJN else None
        pk = ipclPublicKey(p * q, n_length, enable_DJN, hs=hs)
        return pk, ipclPrivateKey(pk, p, q)


class hybridMode(enum.IntEnum):
    """ipcl::HybridMode values (bindings/ipcl_bindings.cpp:37-51); inert here (no QAT on this platform)."""

    OPTIMAL = 0
    QAT = 1
    PREF_QAT90 = 2
    PREF_QAT80 = 3
    PREF_QAT70 = 4
    PREF_QAT60 = 5
    HALF = 6
    PREF_IPP60 = 7
    PREF_IPP70 = 8
    PREF_IPP80 = 9
    PREF_IPP90 = 10
    IPP = 11
    UNDEFINED = 12


class context:
    """QAT context shims (bindings/include/ipcl_bindings.hpp:27-35): accepted and ignored."""

    @staticmethod
    def initializeContext(runtime_choice: str = "") -> bool:
        return True

    @staticmethod
    def terminateContext() -> bool:
        return True

    @staticmethod
    def isQATRunning() -> bool:
        return False

    @staticmethod
    def isQATActive() -> bool:
        return False


class hybridControl:
    _mode = hybridMode.UNDEFINED

    @staticmethod
    def setHybridMode(mode: hybridMode) -> None:
        hybridControl._mode = hybridMode(mode)

    @staticmethod
    def setHybridOff() -> None:
        hybridControl._mode = hybridMode.UNDEFINED

    @staticmethod
    def getHybridMode() -> hybridMode:
        return hybridControl._mode
