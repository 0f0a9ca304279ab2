"""Generates tests/golden/api_transcripts.json by running the programs of tests/_transcripts.py
through the REFERENCE's own Python layer — /root/reference/src/ipcl_python/ipcl_python.py and bindings/fixedpoint.py,
imported by path, in the build container only (nothing of the reference travels: the JSON holds outputs only).

What this pins and what it does not.  The reference's Python layer is pure Python over seven classes of its compiled
extension `ipcl_python.bindings.ipcl_bindings`, whose arithmetic (intel/pailliercryptolib@development + IPP-Crypto,
lib/ipcl.cmake:6-7,30-35) is not in /root/reference and cannot be built here.  This script puts a STAND-IN for that
extension into sys.modules: the same seven classes holding Python ints, with the container semantics the bindings
show (bindings/ipcl_bindings_classes.cpp:165-491: element / slice access with step 1, broadcast of a length-1 right
operand, rotate, getTexts, little-endian bytes padded to 4) and the textbook Paillier maths of SURVEY App. D for the
five forwarded calls (encrypt, apply_obfuscator, decrypt, ct + ct, ct * pt), plus `gmpy2.invert` as `pow(x, -1, m)`.
Obfuscator randomness is drawn from a queue this script fills, so that ciphertext bits are defined.

  => PINNED to the reference's literal code: every COMPOSITION rule of ipcl_python.py — which operand is raised by
     2^delta, result exponents, negative multipliers through the inverted ciphertext, scalar broadcast, `a - b`, `a / s`,
     `@` / `r@` / `@=` index maps and per-row alignment, the padded rotate-and-add reduction, int-vs-float decoding.
  => NOT pinned (nothing can pin it here): the primitives' bits — they are the stand-in's, i.e. mathematics.
     The stand-in decrypts with the lambda/mu formula, the oracle and the device with CRT: two routes to the same m.

    python tests/golden/make_api_transcripts.py
"""
import importlib
import json
import math
import sys
import types
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
REF = Path("/root/reference/src/ipcl_python")

from tests import _transcripts as T  # noqa: E402

R_QUEUE: list = []          # obfuscator randomness consumed by the stand-in, one value per obfuscated element


# ---- stand-in for ipcl_python.bindings.ipcl_bindings (Python ints; semantics: SURVEY App. A / App. D) -----------------
class ipclBigNumber:
    def __init__(self, data=b""):
        self.v = data if isinstance(data, int) else int.from_bytes(bytes(data), "little")

    def to_bytes(self):
        return self.v.to_bytes(max(4, 4 * ((self.v.bit_length() + 31) // 32)), "little")      # BN2bytes: 4-byte padded

    def __eq__(self, other):
        return isinstance(other, ipclBigNumber) and self.v == other.v

    def __hash__(self):
        return hash(self.v)


ipclBigNumber.Zero, ipclBigNumber.One, ipclBigNumber.Two = ipclBigNumber(0), ipclBigNumber(1), ipclBigNumber(2)


def _bn_list(data):
    if isinstance(data, ipclBigNumber):
        return [data.v]
    return [b.v for b in data]


class ipclPlainText:
    def __init__(self, data):
        self.t = _bn_list(data)

    def getTexts(self):
        return [ipclBigNumber(v) for v in self.t]

    def __len__(self):
        return len(self.t)


class ipclPublicKey:
    def __init__(self, n, bits, djn, hs=None, randbits=None):
        self._n = n.v if isinstance(n, ipclBigNumber) else int(n)
        self._nsq = self._n * self._n
        self.length, self._djn, self._hs, self._randbits = bits, djn, hs, randbits

    @property
    def n(self):
        return ipclBigNumber(self._n)

    def __eq__(self, other):
        return isinstance(other, ipclPublicKey) and self._n == other._n

    def __hash__(self):
        return hash(self._n)

    def _obf(self):
        r = R_QUEUE.pop(0)
        return pow(self._hs, r, self._nsq) if self._djn else pow(r, self._n, self._nsq)

    def encrypt(self, pt, make_secure):
        out = []
        for m in pt.t:
            c = (1 + m * self._n) % self._nsq
            out.append(c * self._obf() % self._nsq if make_secure else c)
        return ipclCipherText(self, [ipclBigNumber(c) for c in out])

    def apply_obfuscator(self, x):
        if isinstance(x, ipclBigNumber):
            return ipclBigNumber(x.v * self._obf() % self._nsq)
        return [ipclBigNumber(c * self._obf() % self._nsq) for c in x.t]


class ipclCipherText:
    def __init__(self, pk, data):
        self.public_key, self.t = pk, _bn_list(data)

    def __len__(self):
        return len(self.t)

    def __getitem__(self, k):
        if isinstance(k, slice):
            start, stop, step = k.indices(len(self.t))
            if step != 1:
                raise RuntimeError("Step size not supported")
            return [ipclBigNumber(v) for v in self.t[start:stop]]
        if not 0 <= int(k) < len(self.t):
            raise IndexError("index out of range")
        return ipclBigNumber(self.t[int(k)])

    def getTexts(self):
        return [ipclBigNumber(v) for v in self.t]

    def rotate(self, shift):
        s = shift % len(self.t)
        return ipclCipherText(self.public_key, [ipclBigNumber(v) for v in self.t[s:] + self.t[:s]])

    def __add__(self, other):
        nsq = self.public_key._nsq
        if len(other.t) == 1:
            return ipclCipherText(self.public_key, [ipclBigNumber(a * other.t[0] % nsq) for a in self.t])
        if len(other.t) != len(self.t):
            raise RuntimeError("CipherText size mismatch")
        return ipclCipherText(self.public_key, [ipclBigNumber(a * b % nsq) for a, b in zip(self.t, other.t)])

    def __mul__(self, pt):
        nsq = self.public_key._nsq
        es = pt.t * len(self.t) if len(pt.t) == 1 else pt.t
        if len(es) != len(self.t):
            raise RuntimeError("PlainText size mismatch")
        return ipclCipherText(self.public_key, [ipclBigNumber(pow(c, e, nsq)) for c, e in zip(self.t, es)])


class ipclPrivateKey:
    def __init__(self, pk, p, q):
        self._pk = pk
        self._p, self._q = sorted((p.v, q.v))
        n = pk._n
        assert self._p * self._q == n
        self._lam = (self._p - 1) * (self._q - 1) // math.gcd(self._p - 1, self._q - 1)
        self._mu = pow((pow(n + 1, self._lam, n * n) - 1) // n, -1, n)

    n = property(lambda self: ipclBigNumber(self._pk._n))
    p = property(lambda self: ipclBigNumber(self._p))
    q = property(lambda self: ipclBigNumber(self._q))

    def decrypt(self, ct):
        n = self._pk._n
        return ipclPlainText([ipclBigNumber((pow(c, self._lam, n * n) - 1) // n * self._mu % n) for c in ct.t])


class ipclKeypair:
    @staticmethod
    def generate_keypair(n_length, enable_DJN):
        raise NotImplementedError("the transcripts use fixed keys")


def _install():
    standin = types.ModuleType("ipcl_python.bindings.ipcl_bindings")
    for cls in (ipclBigNumber, ipclPlainText, ipclCipherText, ipclPublicKey, ipclPrivateKey, ipclKeypair):
        setattr(standin, cls.__name__, cls)
    gmpy2 = types.ModuleType("gmpy2")
    gmpy2.invert = lambda a, m: pow(int(a), -1, int(m))
    pkg = types.ModuleType("ipcl_python")
    pkg.__path__ = [str(REF)]                         # package shell: the reference's __init__ is not executed
    sub = types.ModuleType("ipcl_python.bindings")
    sub.__path__ = [str(REF / "bindings")]            # fixedpoint.py is the reference's own file
    sys.modules.update({"ipcl_python": pkg, "ipcl_python.bindings": sub, "ipcl_python.bindings.ipcl_bindings": standin,
                        "gmpy2": gmpy2})
    return importlib.import_module("ipcl_python.ipcl_python")


class RefBackend:
    """The reference's PaillierPublicKey / PaillierPrivateKey / PaillierEncryptedNumber on the stand-in."""

    def __init__(self, ref, n, p, q, bits, hs, randbits):
        self.randbits = randbits
        self.pk = ref.PaillierPublicKey(ipclPublicKey(n, bits, True, hs=hs, randbits=randbits))
        self.sk = ref.PaillierPrivateKey(self.pk, p, q)

    def enc(self, values, seed):
        import numpy as np
        count = 1 if np.isscalar(values) else len(values)
        assert not R_QUEUE
        R_QUEUE.extend(T.synth_r(seed, count, self.randbits))
        en = self.pk.encrypt(values)
        assert not R_QUEUE
        return en

    def raw(self, values):
        return self.pk.raw_encrypt(values)

    def obf(self, en, seed):
        R_QUEUE.extend(T.synth_r(seed, len(en), self.randbits))
        en.apply_obfuscator()
        assert not R_QUEUE

    def dump(self, en):
        cts = [b.v for b in en.ciphertextBN()]
        dec = self.sk.decrypt(en)
        return en.exponent(), cts, (dec if len(en) > 1 else [dec])


def main():
    ref = _install()
    keys = json.loads((ROOT / "tests" / "golden" / "fixture_keys.json").read_text())
    out = {"_about": "outputs of /root/reference/src/ipcl_python/ipcl_python.py on a Python-int stand-in of its compiled "
                     "bindings; generated by tests/golden/make_api_transcripts.py; programs in tests/_transcripts.py",
           "djn_x": hex(T.DJN_X), "keys": {}}
    for bits in T.KEY_BITS:
        p, q = sorted((int(keys[str(bits)]["p"], 16), int(keys[str(bits)]["q"], 16)))
        n = p * q
        hs = pow((-T.DJN_X * T.DJN_X) % (n * n), n, n * n)
        B = RefBackend(ref, n, p, q, bits, hs, bits // 2)
        progs = {}
        for name in T.PROGRAMS:
            progs[name] = T.run_program(name, B)
            print(bits, name, {k: len(v["ct"]) for k, v in progs[name].items()}, flush=True)
        out["keys"][str(bits)] = progs
    path = ROOT / "tests" / "golden" / "api_transcripts.json"
    path.write_text(json.dumps(out, indent=0, sort_keys=True) + "\n")
    print("wrote", path, path.stat().st_size, "bytes")


if __name__ == "__main__":
    main()
