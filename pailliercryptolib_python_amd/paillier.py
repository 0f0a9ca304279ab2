"""User API: ``PaillierKeypair / PaillierPublicKey / PaillierPrivateKey / PaillierEncryptedNumber / BNUtils``.

Same names, arguments, return conventions and exceptions as ``src/ipcl_python/ipcl_python.py`` of the
reference; each method cites the lines it mirrors.  What differs is underneath:

* a ``PaillierEncryptedNumber`` keeps its ciphertexts as ONE device tensor of limbs and its exponents
  as an int32 numpy array — no per-element Python objects on the hot path (the reference builds an
  ``ipclBigNumber`` per element: ``ipcl_python.py:136-141``);
* exponent alignment (``:570-741``) is one masked device call (``pai_ct_pow2``) instead of Python loops
  over gathered sub-batches;
* negative plaintext multipliers (``:426-437``, ``:470-479``) use the device batch inversion instead of a
  ``gmpy2.invert`` per element — the ciphertext bits are the same ``(ct^-1 mod n^2)^(n - pt)``;
* ``sum()`` implements the evident intent (the reference's raises for len > 1: SURVEY App. B).

Extensions (not in the reference, harmless to its users): ``encrypt(..., r=...)`` to inject the
obfuscator randomness, ``decrypt_to_numpy``, ``PaillierEncryptedNumber.words``.
"""
from __future__ import annotations

import os
from typing import Callable, List, Optional, Tuple, Union

import numpy as np
import torch

from . import bindings as _bindings
from . import engine
from . import fixedpoint as _fp
from .bindings import (
    ipclBigNumber,
    ipclCipherText,
    ipclKeypair,
    ipclPlainText,
    ipclPrivateKey,
    ipclPublicKey,
    merge_taint,
)
from .fixedpoint import FixedPointNumber


class PaillierKeypair:
    @staticmethod
    def generate_keypair(n_length: int = 1024, enable_DJN: bool = True) -> Tuple["PaillierPublicKey", "PaillierPrivateKey"]:
        """ipcl_python.py:20-40."""
        pub, pri = ipclKeypair.generate_keypair(n_length, enable_DJN)
        return PaillierPublicKey(pub), PaillierPrivateKey(pri)


class PaillierPublicKey:
    def __init__(self, key: Union[ipclPublicKey, "PaillierPublicKey", int], n_length: Optional[int] = None,
                 enable_DJN: Optional[bool] = None):
        """ipcl_python.py:44-77 (the copy-constructor branch really copies here; upstream's is a no-op
        that ends in AttributeError — SURVEY App. B)."""
        if isinstance(key, ipclPublicKey):
            self.n = BNUtils.BN2int(key.n)
            self.pubkey = key
        elif isinstance(key, PaillierPublicKey):
            self.n = key.n
            self.pubkey = key.pubkey
        elif isinstance(key, int) and n_length is not None and enable_DJN is not None:
            self.n = key
            self.pubkey = ipclPublicKey(BNUtils.int2BN(self.n), n_length, enable_DJN)
        else:
            raise ValueError(
                "PaillierPublicKey: PubKey should be either key value (n),"
                "PaillierPublicKey or IPP-PaillierPublicKey object"
            )
        self.max_int = self.n // 3 - 1
        self.nsquare = self.n * self.n

    def __getstate__(self):
        return self.pubkey

    def __setstate__(self, state):
        self.pubkey = state
        self.n = BNUtils.BN2int(self.pubkey.n)
        self.max_int = self.n // 3 - 1
        self.nsquare = self.n * self.n

    def __repr__(self):
        return repr(self.pubkey)

    def __eq__(self, other):
        return self.n == other.n

    def __hash__(self):
        return hash(self.pubkey)

    def apply_obfuscator(self, x: Union[int, ipclBigNumber]):
        """ipcl_python.py:97-101."""
        if isinstance(x, int):
            return self.pubkey.apply_obfuscator(BNUtils.int2BN(x))
        return self.pubkey.apply_obfuscator(x)

    def precompute_obfuscators(self, count: int) -> None:
        """Extension: fill the key's obfuscator pool; the next `count` encrypted elements cost one
        modular multiplication each (bindings.ipclPublicKey.fill_obfuscator_pool)."""
        self.pubkey.fill_obfuscator_pool(int(count))

    def raw_encrypt(self, plaintext: Union[np.ndarray, list, int, float]) -> "PaillierEncryptedNumber":
        return self.encrypt(plaintext, apply_obfuscator=False)

    def _encode_plain_addend(self, values, target: np.ndarray):
        """(device residues [N, n_words], exponents int32[N]) of a float batch or an integer ndarray encoded AT the target
        exponents (pai_fp_encode_at: the plaintext side of ct + plaintext, see encrypt's _align_to), or None when the batch
        takes encrypt's general path (other element types, tiny moduli, device lists).  Raises what encrypt raises for NaN / inf."""
        pub = self.pubkey
        is_f64 = _fp.is_float_batch(values) and self.n.bit_length() > 66
        is_i64 = (isinstance(values, np.ndarray) and values.dtype in (np.int16, np.int32, np.int64) and values.ndim == 1
                  and values.shape[0] > 0 and self.n.bit_length() > 66)
        if not (is_f64 or is_i64) or pub.fanout_devices(len(values)) is not None:
            return None
        tgt = np.ascontiguousarray(np.asarray(target, dtype=np.int32).reshape(-1))
        if tgt.shape[0] not in (1, len(values)):
            return None
        h = pub.handle
        x = _fp.checked_float64(values) if is_f64 else np.ascontiguousarray(values, dtype=np.int64)
        # (the host rule first: nothing then stands between the codec launch and the caller's next launch)
        expos = _fp.float64_exponents_at(x, tgt, self.n.bit_length()) if is_f64 and x.shape[0] <= HOST_EXPO_MAX else None
        xs, ts = engine.small_operands([x, tgt], h.device)
        m, expo_d = h.fp_encode_at(xs, ts)
        return m, (expos if expos is not None else expo_d.cpu().numpy())

    def encrypt(self, values: Union[np.ndarray, list, int, float], apply_obfuscator: bool = True, *,
                r: Optional[Union[torch.Tensor, np.ndarray]] = None, _align_to=None) -> "PaillierEncryptedNumber":
        """ipcl_python.py:108-147.  Scalars, lists and 1-D arrays of ints/floats; anything else is a
        ValueError exactly as there (2-D arrays fail the per-element type check).
        _align_to (private; raw encryptions only — the plaintext operand of ct + plaintext): target exponents, one per
        element or a single one.  An element whose own exponent is lower is encoded AT its target: for a raw encryption
        ct^(2^d) = E_raw(m 2^d mod n), so the bits and exponents are those the reference's alignment
        (ipcl_python.py:570-741) produces, without the squarings."""
        if np.isscalar(values):
            values = [values]
        if isinstance(values, np.ndarray):
            ok = values.ndim == 1 and (np.issubdtype(values.dtype, np.integer) or np.issubdtype(values.dtype, np.floating))
            if values.dtype == object:
                ok = values.ndim == 1 and all(isinstance(v, (int, float, np.integer, np.floating)) for v in values)
        else:
            ok = all(isinstance(v, (int, float, np.integer, np.floating)) for v in values)
        if not ok:
            raise ValueError("PaillierPublicKey.encrypt: input value(s) should be integer or float")
        pub = self.pubkey
        h = pub.handle
        if isinstance(r, np.ndarray):
            r = engine.to_device_words(r, h.device)
        tgt = None
        if _align_to is not None and not apply_obfuscator:
            tgt = np.ascontiguousarray(np.asarray(_align_to, dtype=np.int32).reshape(-1))
            if tgt.shape[0] not in (1, len(values)):
                tgt = None
        is_f64 = _fp.is_float_batch(values) and self.n.bit_length() > 66
        is_i64 = (isinstance(values, np.ndarray) and values.dtype in (np.int16, np.int32, np.int64) and values.ndim == 1
                  and values.shape[0] > 0 and self.n.bit_length() > 66)
        devs = pub.fanout_devices(len(values)) if (is_f64 or is_i64) else None
        if apply_obfuscator and r is None and pub.obfuscator_pool_size() >= len(values):
            devs = None                                            # pooled obfuscators live on the home device
        if devs is not None:
            # Multi-GPU: contiguous block shards, each H2D'd straight from the host array to its own
            # device (8 B per element), encoded, obfuscated and encrypted there; ciphertext shards are gathered onto
            # the home device with peer copies.
            x = _fp.checked_float64(values) if is_f64 else np.ascontiguousarray(values, dtype=np.int64)
            r_sh = engine.scatter_shards(r, devs) if (apply_obfuscator and r is not None) else None

            def work(g, dev, begin, count):
                hg = pub.handle_on(dev)
                if count == 0:
                    return hg.empty_ct(0), np.zeros(0, dtype=np.int32)
                xs = torch.from_numpy(x[begin:begin + count]).to(dev)
                if tgt is not None:
                    tg = tgt if tgt.shape[0] == 1 else tgt[begin:begin + count]
                    m, expo_d = hg.fp_encode_at(xs, torch.from_numpy(tg).to(dev))
                    return hg.raw_encrypt(m), expo_d.cpu().numpy()
                m, expo_d = hg.fp_encode_f64(xs) if is_f64 else hg.fp_encode_i64(xs)
                if not apply_obfuscator:
                    ct_g = hg.raw_encrypt(m)
                else:
                    ct_g = hg.encrypt(m, pub._draw_r(count, hg) if r_sh is None else r_sh[g])
                return ct_g, (expo_d.cpu().numpy() if is_f64 else np.zeros(count, dtype=np.int32))

            parts = engine.fan_out(devs, work, len(values))
            ct = engine.gather_shards([p[0] for p in parts], h.device)
            expos = np.concatenate([p[1] for p in parts])
            return PaillierEncryptedNumber(self, ipclCipherText(pub, ct), exponents=expos, length=len(values))
        if (is_f64 or is_i64) and tgt is not None:
            x = _fp.checked_float64(values) if is_f64 else np.ascontiguousarray(values, dtype=np.int64)
            xs, ts = engine.small_operands([x, tgt], h.device)
            m, expo_d = h.fp_encode_at(xs, ts)
            # small float batches: the exponents are a function of the inputs alone — computed here, no read-back (and no
            # synchronisation) between the codec kernel and the encryption
            expos = _fp.float64_exponents_at(x, tgt, self.n.bit_length()) if is_f64 and x.shape[0] <= HOST_EXPO_MAX \
                else expo_d.cpu().numpy()
        elif is_f64:
            # float arrays: 8 B per element cross PCIe and the codec runs on the device (pai_fp_encode_f64)
            x = _fp.checked_float64(values)
            m, expo_d = h.fp_encode_f64(engine.small_operands([x], h.device)[0])
            if x.shape[0] <= HOST_EXPO_MAX:
                # small batches: every launch first — the exponents (a host rule, no read-back) are computed while the device works
                ct = pub.encrypt_words(m, apply_obfuscator, r)
                return PaillierEncryptedNumber(self, ipclCipherText(self.pubkey, ct), exponents=_fp.float64_exponents(x), length=len(values))
            expos = expo_d.cpu().numpy()
        elif is_i64:
            # the integer dtypes the reference's codec accepts (fixedpoint.py:72): exponent 0, residue = x mod n
            m, expo_d = h.fp_encode_i64(torch.from_numpy(np.ascontiguousarray(values, dtype=np.int64)).to(h.device))
            expos = np.zeros(values.shape[0], dtype=np.int32)
        else:
            residues, expos = _fp.encode_array(values, self.n, self.max_int, h.n_words)
            if tgt is not None:
                residues, expos = _fp.align_encoded(residues, expos, tgt, self.n, self.max_int)
            m = engine.to_device_words(residues, h.device)
        ct = pub.encrypt_words(m, apply_obfuscator, r)
        return PaillierEncryptedNumber(self, ipclCipherText(self.pubkey, ct), exponents=expos, length=len(values))


class PaillierPrivateKey:
    def __init__(self, key: Union[ipclPrivateKey, ipclPublicKey, PaillierPublicKey], p: Optional[int] = None,
                 q: Optional[int] = None):
        """ipcl_python.py:151-188."""
        if isinstance(key, ipclPrivateKey):
            self.prikey = key
            self.__n = BNUtils.BN2int(key.n)
            self.__max_int = self.__n // 3 - 1
        elif isinstance(key, ipclPublicKey) and p is not None and q is not None:
            self.prikey = ipclPrivateKey(key, BNUtils.int2BN(p), BNUtils.int2BN(q))
            self.__n = BNUtils.BN2int(key.n)
            self.__max_int = self.__n // 3 - 1
        elif isinstance(key, PaillierPublicKey) and p is not None and q is not None:
            self.prikey = ipclPrivateKey(key.pubkey, BNUtils.int2BN(p), BNUtils.int2BN(q))
            self.__n = key.n
            self.__max_int = key.max_int
        else:
            raise KeyError("PaillierPrivateKey: key should be either Private key or Public key (with p and q)")

    def __getstate__(self):
        return (self.prikey, self.__n, self.__max_int)

    def __setstate__(self, state):
        (self.prikey, self.__n, self.__max_int) = state

    def __eq__(self, other: "PaillierPrivateKey"):
        return (self.prikey.p == other.prikey.p) and (self.prikey.q == other.prikey.q)

    def __hash__(self):
        return hash(self.prikey)

    def __repr__(self):
        return repr(self.prikey)

    def _decrypt_words(self, enc: "PaillierEncryptedNumber") -> np.ndarray:
        # `.words` checks the outcome of the asynchronous inversions behind THIS ciphertext (ipclCipherText._check)
        return engine.to_host_words(self.prikey.decrypt_words(enc.words))

    def _decrypt_mantissas(self, enc: "PaillierEncryptedNumber"):
        """(int64 mantissas, None) through the device decoder, or (None, residue words) when some element
        needs the exact big-integer path (|mantissa| >= 2^63, overflow zone, corrupt residue).  Large batches are
        sharded over the key's devices: ciphertext shards go out by peer copy, every device decrypts and decodes
        its block, and only 8-byte mantissas come back."""
        # one device list for decryption AND decoding: the private key's own (the ciphertext's key object may name other
        # devices; the codec only depends on n, which the two share)
        pub = self.prikey._pk
        home = pub.handle.device
        words = enc.words                                # asynchronous inversions behind this ciphertext report here (_check)
        words = words if words.device == home else words.to(home)
        devs = pub.fanout_devices(len(enc)) if self.__n.bit_length() > 66 else None
        if devs is not None:
            ct_sh = engine.scatter_shards(words, devs)

            def work(g, dev, begin, count):
                if count == 0:
                    return np.zeros(0, dtype=np.int64), False, None
                t = self.prikey.handle_on(dev).decrypt(ct_sh[g])
                mant, flag = pub.handle_on(dev).fp_decode_i64(t)
                return mant.cpu().numpy(), bool(flag.any()), t

            parts = engine.fan_out(devs, work, len(enc))
            if not any(p[1] for p in parts):
                return np.concatenate([p[0] for p in parts]), None
            # some element needs the exact host path: the residues of every shard are already there
            return None, np.concatenate([engine.to_host_words(p[2]) for p in parts if p[2] is not None], axis=0)
        t = self.prikey.decrypt_words(words)
        if self.__n.bit_length() > 66:
            mant, flag = pub.handle.fp_decode_i64(t)
            if not bool(flag.any()):
                return mant.cpu().numpy(), None
        return None, engine.to_host_words(t)

    def raw_decrypt(self, ciphertext: "PaillierEncryptedNumber"):
        """ipcl_python.py:207-217: the raw residues as Python ints (scalar if length 1)."""
        if ciphertext.public_key.n != self.__n:
            raise ValueError("PaillierPrivateKey.raw_decrypt: Public key mismatch")
        ret = engine.words_to_ints(self._decrypt_words(ciphertext))
        return ret if len(ciphertext) > 1 else ret[0]

    def decrypt(self, encrypted_number: "PaillierEncryptedNumber"):
        """ipcl_python.py:219-245: list of decoded values (int when the exponent is <= 0, float otherwise),
        or the single value when the length is 1."""
        if encrypted_number.public_key.n != self.__n:
            raise ValueError("PailierPrivateKey.decrypt: Public key mismatch")
        mant, words = self._decrypt_mantissas(encrypted_number)
        if mant is not None:
            ret = _fp.decode_mantissas(mant, encrypted_number._expo)
        else:
            ret = _fp.decode_array(words, encrypted_number._expo, self.__n, self.__max_int)
        return ret if len(encrypted_number) > 1 else ret[0]

    def decrypt_to_numpy(self, encrypted_number: "PaillierEncryptedNumber") -> np.ndarray:
        """Extension: the decoded values as a float64 ndarray without per-element Python objects."""
        if encrypted_number.public_key.n != self.__n:
            raise ValueError("PailierPrivateKey.decrypt: Public key mismatch")
        mant, words = self._decrypt_mantissas(encrypted_number)
        if mant is not None:
            return np.ldexp(mant.astype(np.float64), -np.asarray(encrypted_number._expo, dtype=np.int64).astype(np.int32))
        return _fp.decode_float64_array(words, encrypted_number._expo, self.__n, self.__max_int)


# batches from this size on are sorted by |exponent difference| before a fused aligned addition (PAI_ALIGN_SORT_MIN)
ALIGN_SORT_MIN = 1 << 15
# a lazily tagged sum (rows x R^k, bindings.ipclCipherText) is brought back to the wire form once |k| passes this bound
DOM_MAX = 12
HOST_EXPO_MAX = 4096         # float batches up to this size take their exponents from the host codec rule (no device read-back)

# pai_ct_addn (csrc/paillier_capi.hip: pai_ct_addn; kernels_paillier.hpp: RPOW_SPAN): a tile's accumulator passes through domain
# tags between c_lo and c_hi and is brought to dom_out by one product with R^(1 + dom_out - c) from a table of |m| <= 48.
ADDN_RPOW_SPAN = 48
ADDN_ACC_DOM_MAX = 16        # tags of the running accumulator between chunks of add_many (never leave that function)


def _addn_tags_fit(tag0: int, tag: int, k: int, dom_out: int) -> bool:
    c_lo = min(tag0, 1) + (k - 1) * min(tag - 1, 0)
    c_hi = max(tag0, 1) + (k - 1) * max(tag - 1, 0)
    return (abs(2 - tag) <= ADDN_RPOW_SPAN and abs(1 + dom_out - c_lo) <= ADDN_RPOW_SPAN
            and abs(1 + dom_out - c_hi) <= ADDN_RPOW_SPAN)


def _addn_dom_out(tag0: int, tag: int, k: int, last: bool) -> int:
    """The tag a chunk of add_many leaves: the wire form at the end; in between the natural tag tag0 + (k - 1)(tag - 1) — no
    fix-up product: sixteen fresh ciphertexts give -15 — as far as the NEXT chunk (this result + 15 operands at tag 0, or
    retagged to 0) still fits the table."""
    if last:
        return 0
    return max(-ADDN_ACC_DOM_MAX, min(ADDN_ACC_DOM_MAX, tag0 + (k - 1) * (tag - 1)))


def _add_aligned(h: engine.PublicKeyHandle, ta: torch.Tensor, tb: torch.Tensor, delta: np.ndarray, dom: int = 0) -> torch.Tensor:
    """pai_ct_add_aligned on (ta, tb) with host-built exponent differences.  The kernel raises a whole wave tile (16 - 64
    neighbouring elements) as often as its LARGEST |delta| asks; on random floats neighbours differ by 0 ... 6, so a tile
    pays for ~6 squarings where its mean element needs 1.3.  Large batches are therefore processed in the order of
    |delta| inside segments (device argsort + row gathers) and scattered back: the same residues in the same places."""
    n = ta.shape[0]
    delta = np.ascontiguousarray(delta, dtype=np.int32)
    if 0 < n <= engine.HOST_STAGE_MAX // 4:
        # small batches: the shifts are read by the kernel from a pinned slot (no copy on the stream); dom != 0 needs its entry
        # constant BEFORE the shifts are staged (a staged operand belongs to the very next call)
        return h.ct_add_aligned(ta, tb, delta, dom=dom)
    d_dev = torch.from_numpy(delta).to(h.device)
    try:
        sort_min = int(os.environ.get("PAI_ALIGN_SORT_MIN", ALIGN_SORT_MIN))
    except ValueError:
        sort_min = ALIGN_SORT_MIN
    if n < sort_min or tb.shape[0] != n or delta.size == 0 or int(np.abs(delta).max()) <= 1:
        return h.ct_add_aligned(ta, tb, d_dev, dom=dom)
    # sort inside segments of 512 neighbours only: the kernel hands every wave one contiguous run of tiles (about that long at
    # 2^20 elements), so a global sort would pile the expensive tiles onto the last waves
    seg = 512
    key = (torch.arange(n, device=h.device, dtype=torch.int64) // seg) * 128 + d_dev.abs().clamp(max=127).to(torch.int64)
    order = torch.argsort(key, stable=True)
    res_s = h.ct_add_aligned(ta.index_select(0, order), tb.index_select(0, order), d_dev.index_select(0, order).contiguous(), dom=dom)
    res = torch.empty_like(res_s)
    res.index_copy_(0, order, res_s)
    return res


class PaillierEncryptedNumber:
    def __init__(self, public_key: PaillierPublicKey, ciphertext: ipclCipherText, exponents, length: int):
        """ipcl_python.py:249-270."""
        if ciphertext.public_key != public_key.pubkey:
            raise ValueError("PaillierEncryptedNumber: public key mismatch")
        self._expo = np.asarray(exponents, dtype=np.int32).reshape(-1).copy()
        self.public_key = public_key
        self.__ipclCipherText = ciphertext
        self.__length = int(length)

    # -- helpers ----------------------------------------------------------------------------------
    @property
    def words(self) -> torch.Tensor:
        """Device limb matrix [N, ct_words] (extension)."""
        return self.__ipclCipherText.words

    @property
    def _w(self) -> torch.Tensor:
        """The wire-form limb matrix for operations that stay on the device: no look at pending inversion outcomes (they
        travel on with the result: _wrap), hence no synchronisation."""
        return self.__ipclCipherText._t

    def _h(self) -> engine.PublicKeyHandle:
        return self.public_key.pubkey.handle

    def _wrap(self, ct: torch.Tensor, expo, length: Optional[int] = None, dom: int = 0, others=(), flags=()) -> "PaillierEncryptedNumber":
        """A result computed from self (and `others`): it inherits their pending inversion outcomes plus the new `flags`."""
        taint = merge_taint(self.__ipclCipherText._taint, *[o.ciphertext()._taint for o in others], tuple(flags))
        return PaillierEncryptedNumber(self.public_key, ipclCipherText(self.public_key.pubkey, ct, dom=dom, taint=taint), expo,
                                       ct.shape[0] if length is None else length)

    def __repr__(self):
        return repr(self.__ipclCipherText)

    def __getstate__(self) -> tuple:
        """ipcl_python.py:281-287: (public_key, len, exponents, [python ints])."""
        return (self.public_key, len(self), self.exponent(), [BNUtils.BN2int(i) for i in self.ciphertextBN()])

    def __setstate__(self, state: tuple):
        (self.public_key, self.__length, expo, ciphertextPyInt) = state
        self._expo = np.asarray(expo, dtype=np.int32).reshape(-1).copy()
        self.__ipclCipherText = ipclCipherText(self.public_key.pubkey, [int(i) for i in ciphertextPyInt])

    def __len__(self) -> int:
        return self.__length

    def ciphertext(self) -> ipclCipherText:
        return self.__ipclCipherText

    def ciphertextBN(self, idx: Optional[int] = None):
        """ipcl_python.py:306-322."""
        if idx is None:
            return self.__ipclCipherText.getTexts()
        if not 0 <= idx < self.__length:
            raise IndexError("ciphertext: idx out of range")
        return self.__ipclCipherText[idx]

    def exponent(self, idx: Optional[int] = None):
        """ipcl_python.py:324-340."""
        if idx is None:
            return [int(e) for e in self._expo]
        if not 0 <= idx < self.__length:
            raise IndexError("exponent: idx out of range")
        return int(self._expo[idx])

    def apply_obfuscator(self, *, r: Optional[torch.Tensor] = None):
        """ipcl_python.py:342-346: re-randomise in place."""
        h = self._h()
        ct = self._w.clone()
        h.obfuscate_(ct, self.public_key.pubkey._draw_r(ct.shape[0]) if r is None else r)
        self.__ipclCipherText = ipclCipherText(self.public_key.pubkey, ct, taint=self.__ipclCipherText._taint)

    def __getitem__(self, key: Union[int, slice]) -> "PaillierEncryptedNumber":
        """ipcl_python.py:348-360 (open-ended slices are accepted as an extension)."""
        if isinstance(key, (int, np.integer)):
            key = slice(int(key), int(key) + 1)
        start = 0 if key.start is None else key.start
        stop = len(self) if key.stop is None else key.stop
        if not 0 <= stop <= len(self) or not 0 <= start < len(self):
            raise IndexError("__getitem__: key out of range")
        if key.step not in (None, 1):
            raise RuntimeError("Step size not supported")
        t, dom = self.__ipclCipherText._raw()                    # a slice keeps its container's domain tag
        return self._wrap(t[start:stop].contiguous(), self._expo[start:stop], dom=dom)

    def __iter__(self):
        return (self[i] for i in range(len(self)))

    # -- arithmetic -------------------------------------------------------------------------------
    def __add__(self, other):
        """ipcl_python.py:365-375."""
        if self.__length == 1 and isinstance(other, PaillierEncryptedNumber) and len(other) > 1:
            return other.__raw_add(self)
        return self.__raw_add(other)

    def __radd__(self, other):
        return self + other

    def __sub__(self, other):
        """ipcl_python.py:383-389: self + other * -1.0."""
        if isinstance(other, list):
            other = np.array(other)
        if (isinstance(other, PaillierEncryptedNumber) and self.public_key == other.public_key
                and (len(other) == self.__length or len(other) == 1) and self.public_key.n.bit_length() > 66):
            return self.__sub_ct(other)
        return self.__raw_add(other * -1.0)

    def __sub_ct(self, other: "PaillierEncryptedNumber") -> "PaillierEncryptedNumber":
        """ct - ct with the reference's bits at half the squarings.  The reference forms nb = (b^-1)^(2^52) (the
        multiplier -1.0 encodes as n - 2^52 with exponent 52: ipcl_python.py:426-437), then aligns: with E =
        max(e_a, e_b + 52) it returns a^(2^(E - e_a)) * nb^(2^(E - e_b - 52)).  Powers of two distribute over the
        product, so the same residue is t^(2^(E - m)) with t = a^(2^(m - e_a)) * (b^-1)^(2^(m - e_b)), m = max(e_a, e_b):
        |e_a - e_b| + (E - m) squarings instead of (E - e_a) + 52 + (E - e_b - 52)."""
        h = self._h()
        xe = np.asarray(self._expo, dtype=np.int64)
        ye = np.asarray(other._expo, dtype=np.int64)
        if other._w.shape[0] == 1 and self._w.shape[0] > 1:
            ye = np.broadcast_to(ye, xe.shape)
        m = np.maximum(xe, ye)
        E = np.maximum(xe, ye + FixedPointNumber.FLOAT_MANTISSA_BITS - 1)
        flag = h.new_flag()
        b_inv = h.ct_invert(other._w, flag=flag)
        t = _add_aligned(h, self._w, b_inv, (xe - ye).astype(np.int32))
        k = (E - m).astype(np.int32)
        if (k > 0).any():
            h.ct_pow2_(t, k)
        return self._wrap(t, E.astype(np.int32), self.__length, others=(other,), flags=(flag,))

    def __rsub__(self, other):
        """ipcl_python.py:391-397: (self * -1.0) + other; for a plaintext `other` the sum raw-encrypts it first (:495-504),
        and the product of the two ciphertexts is commutative, so this is (raw-encrypted other) - self."""
        if isinstance(other, PaillierEncryptedNumber):
            return other - self
        if isinstance(other, list):
            other = np.array(other)
        plain_ok = (isinstance(other, np.ndarray) and other.ndim == 1 and len(other) == self.__length) or \
                   (np.isscalar(other) and isinstance(other, (int, float, np.integer, np.floating)))
        if plain_ok and self.public_key.n.bit_length() > 66:
            pos = self.public_key.encrypt(other, apply_obfuscator=False)
            if len(pos) == self.__length:
                return pos.__sub_ct(self)
        return (self * (-1.0)).__raw_add(other)

    def __rmul__(self, other):
        return self * other

    def __truediv__(self, other):
        """ipcl_python.py:404-410."""
        if isinstance(other, list):
            other = np.array(other)
        inv_other = 1.0 / other
        return self * inv_other

    def _pow(self, ct: torch.Tensor, pts: List[int], neg: np.ndarray, flags: list) -> torch.Tensor:
        """ct_i^(pt_i) with the reference's sign rule: pt >= n - max_int means a negative multiplier, which is
        applied as (ct^-1)^(n - pt) (ipcl_python.py:426-437, 470-479)."""
        h = self._h()
        n = self.public_key.n
        N = ct.shape[0]
        bcast = len(pts) == 1 and N > 1
        if neg.any():
            flags.append(h.new_flag())                   # the inversion's outcome travels with the result (_wrap)
            if bcast or neg.all():
                base = h.ct_invert(ct, flag=flags[-1])
            else:
                idx = torch.from_numpy(np.nonzero(neg)[0]).to(h.device)
                base = ct.clone()
                base[idx] = h.ct_invert(ct[idx].contiguous(), flag=flags[-1])
        else:
            base = ct
        mags = [n - p if s else p for p, s in zip(pts, neg)]
        bits = max(1, max(v.bit_length() for v in mags))
        ew = (bits + 31) // 32
        return self.public_key.pubkey.ct_mul_words(base, engine.ints_to_words(mags, ew), bits)

    def _pow_small(self, ct: torch.Tensor, mant: np.ndarray, flags: list) -> torch.Tensor:
        """The same for a batch of signed 64-bit multipliers (float mantissas): no per-element Python objects."""
        h = self._h()
        neg = mant < 0
        if neg.any():
            flags.append(h.new_flag())
            if neg.all():
                base = h.ct_invert(ct, flag=flags[-1])
            else:
                idx = torch.from_numpy(np.nonzero(neg)[0]).to(h.device)
                base = ct.clone()
                base[idx] = h.ct_invert(ct[idx].contiguous(), flag=flags[-1])
        else:
            base = ct
        mag = np.abs(mant)                               # |mantissa| < 2^53: the int64 bit pattern IS the little-endian word pair
        bits = max(1, int(mag.max()).bit_length())
        e = mag.view(np.uint32).reshape(-1, 2)
        ew = (bits + 31) // 32
        return self.public_key.pubkey.ct_mul_words(base, e if ew == 2 else np.ascontiguousarray(e[:, :ew]), bits)

    def __mul__(self, other):
        """ipcl_python.py:412-488."""
        n, max_int = self.public_key.n, self.public_key.max_int
        if np.isscalar(other):
            enc = FixedPointNumber.encode(other, n, max_int)
            pt, pt_exponent = enc.encoding, enc.exponent
            if not 0 <= pt < n:
                raise ValueError(f"PaillierEncryptedNumber.__mul__: Scalar out ofbounds: {pt}")
            res_expo = self._expo + np.int32(pt_exponent)
            flags: list = []
            res = self._pow(self._w, [pt], np.array([pt >= n - max_int]), flags)
            return self._wrap(res, res_expo, self.__length, flags=flags)
        if len(other) != self.__length:
            raise ValueError("PaillierEncryptedNumber.__mul__: Multiply size mismatch")
        h = self._h()
        flags = []
        if _fp.is_float_batch(other) and n.bit_length() > 66:
            mant, pexpo = _fp.float64_mantissas(_fp.checked_float64(other))
            return self._wrap(self._pow_small(self._w, mant, flags), self._expo + pexpo, self.__length, flags=flags)
        residues, pexpo = _fp.encode_array(other, n, max_int, h.n_words)
        pts = engine.words_to_ints(residues)
        neg = np.array([p >= n - max_int for p in pts])
        res_expo = self._expo + pexpo
        res = self._pow(self._w, pts, neg, flags)
        return self._wrap(res, res_expo, self.__length, flags=flags)

    def __raw_add(self, other):
        """ipcl_python.py:490-526."""
        if isinstance(other, (np.ndarray, list)):
            if self.__length != len(other):
                raise ValueError("PaillierEncryptedNumber.__raw_add: array(list) size mismatch with PaillierEncryptedNumber")
            # the plaintext side is aligned in the plaintext domain (PaillierPublicKey.encrypt: _align_to)
            if 0 < self.__length <= _bindings.EAGER_ADD_MAX and self.__ipclCipherText._raw()[1] == 0:
                # small batches: encode at the ciphertext's exponents and, when that aligned every element (the rule — an element
                # that could not be shifted that far is the exception), ct * (1 + m n) in ONE pass (pai_ct_add_plain) instead
                # of a raw encryption, an intermediate array and an addition
                enc = self.public_key._encode_plain_addend(other, self._expo)
                if enc is not None:
                    m_dev, ye = enc
                    if np.array_equal(ye, self._expo):
                        res = self._h().ct_add_plain(self.__ipclCipherText._raw()[0], m_dev)
                        return self._wrap(res, self._expo, self.__length, dom=0)
                    other = PaillierEncryptedNumber(self.public_key, ipclCipherText(self.public_key.pubkey, self._h().raw_encrypt(m_dev)),
                                                    exponents=ye, length=self.__length)
            if not isinstance(other, PaillierEncryptedNumber):
                other = self.public_key.encrypt(other, apply_obfuscator=False, _align_to=self._expo)
        elif np.isscalar(other) and isinstance(other, (int, float, np.integer, np.floating)):
            other = self.public_key.encrypt(other, apply_obfuscator=False,
                                            _align_to=[int(self._expo.min())] if self.__length else None)
        elif isinstance(other, PaillierEncryptedNumber):
            if self.public_key != other.public_key:
                raise ValueError("PaillierEncryptedNumber.__raw_add: PublicKey mismatch")
            if self.__length != len(other) and len(other) > 1:
                raise ValueError("PaillierEncryptedNumber.__raw_add: CipherText size mismatch with PaillierEncryptedNumber")
        else:
            raise TypeError(f"PaillierEncryptedNumber.__raw_add: unsupported operand {type(other)}")
        # alignment (ipcl_python.py:570-741: the lower-exponent side is raised by ct^(2^delta)) fused with the addition
        h = self._h()
        xe = np.asarray(self._expo, dtype=np.int64)
        ye = np.asarray(other._expo, dtype=np.int64)
        # lazy Montgomery domain (bindings.ipclCipherText._raw): operands hold x R^k; the sum is ONE product and its tag
        # remembers the stray R^-1 — whoever needs the wire form pays the second product once, sums of sums never do.
        # (_raw() first: `.words` would canonicalise the operand — one product and a new buffer — just to read a shape.)
        (ta, ka), (tb, kb) = self.__ipclCipherText._raw(), other.ciphertext()._raw()
        if tb.shape[0] == 1 and ta.shape[0] > 1:
            ye = np.broadcast_to(ye, xe.shape)
        delta = (xe - ye).astype(np.int32)
        if not delta.any():
            if ka == 0 and kb == 0 and ta.shape[0] <= _bindings.eager_ctct_max(self.public_key.pubkey._bits):
                # small batches run on the latency geometry, where the tagged product costs two products anyway (its result
                # has to carry the throughput geometry's R): the wire form straight away — same kernel time, and no retag
                # product (a second launch) when the result is exported, decrypted or multiplied
                return self._wrap(h.ct_add(ta, tb), np.maximum(xe, ye).astype(np.int32), self.__length, dom=0, others=(other,))
            res, dom = h.ct_mont_mul(ta, tb), ka + kb - 1
            if abs(dom) > DOM_MAX:                       # long-lived accumulators: the tag (and the R^k constants it needs) stay bounded
                res, dom = h.ct_retag(res, dom, 0), 0
            return self._wrap(res, np.maximum(xe, ye).astype(np.int32), self.__length, dom=dom, others=(other,))
        if kb != ka:
            tb = h.ct_retag(tb, kb, ka)
        res = _add_aligned(h, ta, tb, delta, dom=ka)
        return self._wrap(res, np.maximum(xe, ye).astype(np.int32), self.__length, dom=ka, others=(other,))

    def increase_exponent_to(self, x_ct, x_expo, exponent: int):
        """ipcl_python.py:528-568: raise every element of x to `exponent` (ct^(2^delta) where delta > 0)."""
        words = x_ct._t if isinstance(x_ct, ipclCipherText) else x_ct
        delta = (np.int64(exponent) - np.asarray(x_expo, dtype=np.int64)).astype(np.int32)
        if (delta > 0).any():
            h = self._h()
            words = words.clone()
            h.ct_pow2_(words, delta)
        return ipclCipherText(self.public_key.pubkey, words, taint=x_ct._taint) if isinstance(x_ct, ipclCipherText) else words

    def __align_exponent(self, x_ct: torch.Tensor, x_expo, y_ct: torch.Tensor, y_expo):
        """ipcl_python.py:570-741: per element, the side with the smaller exponent is multiplied by
        2^delta (as ciphertext^(2^delta)); a length-1 y broadcasts."""
        h = self._h()
        xe = np.asarray(x_expo, dtype=np.int64)
        ye = np.asarray(y_expo, dtype=np.int64)
        if y_ct.shape[0] == 1 and x_ct.shape[0] > 1:
            ye = np.broadcast_to(ye, xe.shape)
        res = np.maximum(xe, ye)
        dx = (res - xe).astype(np.int32)
        dy = (res - ye).astype(np.int32)
        if (dx > 0).any():
            x_ct = x_ct.clone()
            h.ct_pow2_(x_ct, dx)
        if (dy > 0).any():
            if y_ct.shape[0] == 1 and x_ct.shape[0] > 1:
                y_ct = y_ct.expand(x_ct.shape[0], -1).contiguous()
            else:
                y_ct = y_ct.clone()
            h.ct_pow2_(y_ct, dy)
        return x_ct, y_ct, res.astype(np.int32)

    def length(self) -> int:
        return self.__length

    # -- reductions -----------------------------------------------------------------
    def _tree_product(self, ct: torch.Tensor, groups: int) -> torch.Tensor:
        """[members * groups, W] read member-major -> [groups, W]: out[g] = product over l of ct[l * groups + g]
        modulo n^2 (pai_ct_prod: a product tree over halves).  The result does not depend on the association order,
        so it equals the reference's pad-with-E(0)=1-and-rotate scheme (ipcl_python.py:810-827) bit for bit."""
        return self._h().ct_prod(ct.contiguous(), groups)

    def _aligned_tree(self, ct: torch.Tensor, expo: np.ndarray, groups: int):
        """[members * groups, W] read member-major with per-element exponents -> ([groups, W], exponents [groups]):
        out[g] = prod_l ct[l * groups + g]^(2^(max_g - e[l * groups + g])) mod n^2 — what the reference computes by
        raising every element to its group's maximum exponent (ipcl_python.py:746-750, 868-870) and reducing
        (:810-827).  Here the alignment rides along the product tree: a node is pai_ct_add_aligned of its two children
        (the lower-exponent child is raised by the DIFFERENCE of the children's exponents, a few squarings) and carries
        the larger exponent upwards, so every leaf is raised by max_g - e in total along its path — the same residue —
        without a pass that squares every element up to the global maximum first."""
        h = self._h()
        W = ct.shape[1]
        e = np.asarray(expo, dtype=np.int64).reshape(-1, groups)
        if (e == e[0:1]).all():
            return h.ct_prod(ct.contiguous(), groups), e.max(axis=0)          # nothing to align: one product per node
        x = ct.reshape(-1, groups, W)
        while x.shape[0] > 1:
            n = x.shape[0]
            hh = (n + 1) // 2
            lo = n - hh
            ea, eb = e[:lo], e[hh:hh + lo]
            prod = _add_aligned(h, x[:lo].reshape(-1, W).contiguous(), x[hh:hh + lo].reshape(-1, W).contiguous(),
                                (ea - eb).reshape(-1).astype(np.int32))
            prod = prod.reshape(lo, groups, W)
            em = np.maximum(ea, eb)
            if lo < hh:
                x = torch.cat([prod, x[lo:hh]], dim=0)
                e = np.concatenate([em, e[lo:hh]], axis=0)
            else:
                x, e = prod, em
        return x.reshape(groups, W).contiguous(), e.reshape(groups)

    # n-ary sums of at least this many elements per operand go through pai_ct_addn (below: the pairwise kernels, which serve
    # small batches on the latency geometry)
    ADDN_MIN = 4096

    @staticmethod
    def add_many(items) -> "PaillierEncryptedNumber":
        """Extension: items[0] + items[1] + ... + items[k-1] for equally long ciphertext arrays of one key — the aggregation the
        reference spells as a chain of __add__ (ipcl_python.py:365-381, 490-526; tests/ipcl_python_test.py:21-38) — in ONE
        pass per 16 operands (pai_ct_addn: k - 1 products per element, one store, no intermediate arrays, the result in the
        wire form).  Exponents are aligned to the per-element maximum as the chain does (:570-741), so ciphertext bits and
        exponents equal those of the chain.  Operand sets whose alignment would cost more inside the one-pass kernel than in
        the pairwise kernels (exponents that differ by more than a squaring per operand on average), short arrays and
        broadcasts take the chain itself."""
        items = list(items)
        if not items:
            raise ValueError("PaillierEncryptedNumber.add_many: nothing to add")
        first = items[0]
        if not all(isinstance(x, PaillierEncryptedNumber) for x in items):
            raise TypeError("PaillierEncryptedNumber.add_many: operands must be PaillierEncryptedNumber")
        k, n = len(items), len(first)
        same = all(x.public_key == first.public_key and len(x) == n for x in items)

        def chain():
            acc = items[0]
            for x in items[1:]:
                acc = acc + x
            return acc

        if k < 3 or not same or n < PaillierEncryptedNumber.ADDN_MIN:
            return chain()
        E = items[0]._expo.astype(np.int64)
        for x in items[1:]:
            E = np.maximum(E, x._expo)
        raises = [(E - x._expo).astype(np.int32) for x in items]
        peak = [int(r.max()) for r in raises]
        # one-pass cost of the alignment: a domain entry + max(raise) squarings per raised operand and tile
        if sum(1 + p for p in peak if p > 0) > 2 * (k - 1):
            return chain()
        h = first._h()
        acc_t, acc_tag, acc_raise = None, 0, None
        pos = 0
        while pos < k:
            take = items[pos:pos + (16 if acc_t is None else 15)]
            ops, tags, rz = [], [], []
            for x, r, p in zip(take, raises[pos:pos + len(take)], peak[pos:pos + len(take)]):
                t, tag = x.ciphertext()._raw()
                ops.append(t)
                tags.append(tag)
                rz.append(torch.from_numpy(r).to(h.device) if p > 0 else None)
            pos += len(take)
            # operands 1.. share one tag (the most common one; the others are retagged: one product each, rare)
            rest = tags if acc_t is not None else tags[1:]
            tag = max(set(rest), key=rest.count) if rest else 0
            for i in range(len(ops)):
                own_first = acc_t is None and i == 0
                if tags[i] != tag and not (own_first and rz[0] is None):
                    ops[i] = h.ct_retag(ops[i], tags[i], tag)
                    tags[i] = tag
            if acc_t is not None:
                ops, rz, tag0 = [acc_t] + ops, [None] + rz, acc_tag
            else:
                tag0 = tags[0]
            last = pos >= k
            dom_out = _addn_dom_out(tag0, tag, len(ops), last)
            if not _addn_tags_fit(tag0, tag, len(ops), dom_out):
                # operands that each carry a long lazy chain (e.g. sixteen sums a + b + c + d at tag -3): the fix-up constant
                # R^(1 + dom_out - c) would leave the key's table (pai_ct_addn: RPOW_SPAN) — bring operands 1.. to the wire
                # form first (one product each), which always fits (tests/test_host_logic_cpu.py::test_addn_tag_plan)
                # (operand 0 is the running sum or the first item; everything after it sits at `tag` by now)
                if tag != 0:
                    for i in range(1, len(ops)):
                        ops[i] = h.ct_retag(ops[i], tag, 0)
                if acc_t is None and rz[0] is not None and tag0 != 0:     # a raised first operand shares the others' tag
                    ops[0] = h.ct_retag(ops[0], tag0, 0)
                    tag0 = 0
                tag = 0
                dom_out = _addn_dom_out(tag0, tag, len(ops), last)
            acc_t = h.ct_addn(ops, rz, tag0, tag, dom_out)
            acc_tag = dom_out
        return first._wrap(acc_t, E.astype(np.int32), n, dom=acc_tag, others=items[1:])

    def sum(self) -> "PaillierEncryptedNumber":
        """ipcl_python.py:746-762 (intended behaviour)."""
        out, e = self._aligned_tree(self._w, self._expo, 1)
        return self._wrap(out, [int(e[0])], 1)

    def mean(self) -> "PaillierEncryptedNumber":
        return self.sum() / len(self)

    def dot(self, other: Union[np.ndarray, list]) -> "PaillierEncryptedNumber":
        if len(other) != len(self):
            raise ValueError("PaillierEncryptedNumber.dot: input size mismatch with ciphertext")
        fast = self._matmul_multiexp(np.asarray(other), 1, len(self), 1, False) if _fp.is_float_batch(other) else None
        if fast is not None:
            return fast
        return (self * other).sum()

    # matrix products with at least this many ciphertext * plaintext terms go through pai_ct_multiexp (PAI_MEXP_MIN_TERMS)
    MEXP_MIN_TERMS = 1 << 19

    def _matmul_multiexp(self, other: np.ndarray, m: int, n: int, k: int, rhs: bool):
        """The same output as __matmul below for float matrices, as ONE multi-exponentiation per output element
        (pai_ct_multiexp): out(i, j) = prod_l base^(|mantissa| << shift), the shift being the exponent alignment of
        ipcl_python.py:868-870 and the base the ciphertext or — negative multipliers, :426-437 — its inverse.  The bits
        are those of the term-by-term route (a canonical residue of the same product).  Returns None when the shape,
        the key or the exponent spread are not served; the caller then goes term by term."""
        import os

        from . import _native
        h = self._h()
        try:
            min_terms = int(os.environ.get("PAI_MEXP_MIN_TERMS", self.MEXP_MIN_TERMS))
        except ValueError:
            min_terms = self.MEXP_MIN_TERMS
        if m * n * k < min_terms or other.dtype.kind != "f" or self.public_key.n.bit_length() <= 66:
            return None
        dev = h.device
        # the limits pai_ct_multiexp enforces (terms < 2^31, bases < 2^28) and a memory estimate are checked BEFORE the
        # per-term exponent tensors are built (~80 bytes per term live at the peak): a shape that cannot be served goes
        # term by term instead of dying in the allocator
        terms = m * n * k
        if terms >= 1 << 31 or (k * n if rhs else m * n) >= 1 << 28:
            return None
        if dev.type == "cuda" and 96 * terms > torch.cuda.mem_get_info(dev)[0]:
            return None
        try:
            return self._matmul_multiexp_build(other, m, n, k, rhs, h)
        except torch.cuda.OutOfMemoryError:
            return None

    def _matmul_multiexp_build(self, other: np.ndarray, m: int, n: int, k: int, rhs: bool, h):
        from . import _native
        dev = h.device
        if rhs:
            # out(i, j) = sum_l other[i, l] * self[l*k + j]: rows of bases <-> j, members <-> l, columns <-> i
            R, K, M = k, n, m
            idx = (np.arange(n)[None, :] * k + np.arange(k)[:, None]).reshape(-1)             # [j][l] -> l*k + j
            bases = self._w[torch.from_numpy(idx).to(dev)].contiguous()
            e_a = np.asarray(self._expo, dtype=np.int64)[idx].reshape(R, K)
            w = (other.T if other.ndim == 2 else other.reshape(1, n).T)                         # [l][i]
        else:
            R, K, M = m, n, k
            bases = self._w
            e_a = np.asarray(self._expo, dtype=np.int64).reshape(R, K)
            w = other if other.ndim == 2 else other.reshape(n, 1)                               # [l][j]
        mant, pexpo = _fp.float64_mantissas(_fp.checked_float64(np.ascontiguousarray(w, dtype=np.float64).reshape(-1)))
        mant_t = torch.from_numpy(mant.reshape(K, M)).to(dev)
        total = torch.from_numpy(e_a).to(dev)[:, :, None] + torch.from_numpy(pexpo.astype(np.int64).reshape(K, M)).to(dev)[None]
        big_e = total.amax(dim=1)                                                               # [R, M]
        mag = mant_t.abs()
        shift = torch.where(mag[None] == 0, torch.zeros_like(total), big_e[:, None, :] - total)
        del total
        bitlen = torch.frexp(mag.to(torch.float64))[1].to(torch.int64)                          # exact below 2^53
        ebits = int((bitlen[None] + shift).max().item())
        if ebits > 128:
            return None
        ebits = max(ebits, 1)
        ew = (ebits + 31) // 32
        m32 = 0xFFFFFFFF
        mag_b = mag[None].expand_as(shift)
        words = []
        for wi in range(ew):
            pos = 32 * wi - shift                                                               # bit of mag at the word's bit 0
            right = (mag_b >> pos.clamp(0, 63)) & m32
            left = ((mag_b & m32) << (-pos).clamp(0, 31)) & m32
            words.append(torch.where(pos >= 0, right, torch.where(pos > -32, left, torch.zeros_like(left))))
        e_t = torch.stack(words, dim=3)
        e_t = torch.where(e_t >= (1 << 31), e_t - (1 << 32), e_t).to(torch.int32).contiguous()
        del words, shift
        neg = mant_t < 0
        sign = inv = None
        flags = []
        if bool(neg.any().item()):
            flags.append(h.new_flag())
            if M == 1:
                # one column: every base is used with one sign only — swap the inverted ciphertexts in (half the tables)
                rows = torch.nonzero(neg[:, 0].repeat(R)).reshape(-1)
                bases = bases.clone()
                bases[rows] = h.ct_invert(bases[rows].contiguous(), flag=flags[-1])
            else:
                sign = neg.to(torch.uint8).contiguous()
                inv = h.ct_invert(bases, flag=flags[-1])
        try:
            out = h.ct_multiexp(bases, inv, R, K, M, e_t, ebits, sign)
        except _native.NativeError as exc:
            if exc.code == _native.PAI_E_UNSUPPORTED:
                return None
            raise
        expo = big_e.cpu().numpy().astype(np.int32)                                             # [R, M]
        if rhs:
            perm = (np.arange(k)[None, :] * m + np.arange(m)[:, None]).reshape(-1)              # (i, j) <- j*m + i
            out = out[torch.from_numpy(perm).to(dev)].contiguous()
            expo = expo.T
        return self._wrap(out, np.ascontiguousarray(expo).reshape(-1), m * k, flags=flags)

    def __matmul(self, other: np.ndarray, m: int, n: int, k: int, rhs: bool = False) -> "PaillierEncryptedNumber":
        """ipcl_python.py:829-880.  self is (m x n) row-major when rhs is False (result = self @ other,
        other n x k), or (n x k) when rhs is True (result = other @ self, other m x n).  Output element
        (i, j) = sum_l ct[.] * pt[.], one aligned tree product per output element."""
        fast = self._matmul_multiexp(other, m, n, k, rhs)
        if fast is not None:
            return fast
        h = self._h()
        # member-major order (l, i, j): the n addends of output element (i, j) are n rows that lie m*k apart, which is
        # the layout pai_ct_prod reduces with contiguous halves
        l_idx, i_idx, j_idx = np.meshgrid(np.arange(n), np.arange(m), np.arange(k), indexing="ij")
        if rhs:
            idx_self = (l_idx * k + j_idx).reshape(-1)
            pts = (other[i_idx, l_idx] if other.ndim == 2 else other[l_idx]).reshape(-1)
        else:
            idx_self = (i_idx * n + l_idx).reshape(-1)
            pts = (other[l_idx, j_idx] if other.ndim == 2 else other[l_idx]).reshape(-1)
        gather = torch.from_numpy(idx_self).to(h.device)
        big = self._wrap(self._w[gather].contiguous(), self._expo[idx_self], idx_self.shape[0])
        prod = big * np.asarray(pts)
        out, gmax = self._aligned_tree(prod._w, prod._expo, m * k)        # per output element (ipcl_python.py:868-870)
        return self._wrap(out, gmax.astype(np.int32), m * k, others=(prod,))

    def __matmul__(self, other: Union[np.ndarray, list]) -> "PaillierEncryptedNumber":
        """ipcl_python.py:882-903."""
        if len(self) % len(other) != 0:
            raise ValueError("PaillierEncryptedNumber.__matmul__: matrix multiply size mismatch")
        other = np.array(other)
        if other.ndim not in (1, 2):
            raise NotImplementedError(f"PaillierEncryptedNumber.__matmul__: input ndim {other.ndim}not supported")
        n = other.shape[0]
        k = other.shape[1] if other.ndim == 2 else 1
        m = len(self) // n
        return self.__matmul(other, m, n, k)

    def __rmatmul__(self, other: Union[np.ndarray, list]) -> "PaillierEncryptedNumber":
        """ipcl_python.py:905-925."""
        other = np.array(other)
        if other.ndim not in (1, 2):
            raise NotImplementedError(f"PaillierEncryptedNumber.__rmatmul__: input ndim {other.ndim} not supported")
        m = other.shape[0] if other.ndim == 2 else 1
        n = other.shape[1] if other.ndim == 2 else other.shape[0]
        if len(self) % n != 0:
            raise ValueError("PaillierEncryptedNumber.__rmatmul__: matrix multiplysize mismatch")
        k = len(self) // n
        return self.__matmul(other, m, n, k, rhs=True)

    def __imatmul__(self, other: Union[np.ndarray, list]) -> "PaillierEncryptedNumber":
        return self @ other

    # keep numpy from broadcasting `ndarray @ PaillierEncryptedNumber` element-wise
    __array_ufunc__ = None


class BNUtils:
    """ipcl_python.py:933-977."""

    @staticmethod
    def int2Bytes(val: int) -> bytes:
        return val.to_bytes((val.bit_length() + 7) // 8, byteorder="little")

    @staticmethod
    def bytes2Int(val: bytes) -> int:
        return int.from_bytes(val, "little")

    @staticmethod
    def int2BN(val: int) -> ipclBigNumber:
        if val == 0:
            return ipclBigNumber.Zero
        if val == 1:
            return ipclBigNumber.One
        if val == 2:
            return ipclBigNumber.Two
        return ipclBigNumber(BNUtils.int2Bytes(val))

    @staticmethod
    def BN2int(val: ipclBigNumber) -> int:
        return BNUtils.bytes2Int(val.to_bytes())
