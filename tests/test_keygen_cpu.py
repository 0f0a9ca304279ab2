"""CPU: the native key generator (pai_keygen — ipcl::generateKeypair, bindings/ipcl_bindings.cpp:12-15) and the host
modexp used for the DJN base (pai_host_modexp) against CPython integers.  Host-only entry points: no GPU needed."""
import math
import secrets

import pytest


@pytest.fixture(scope="module")
def native():
    from pailliercryptolib_python_amd import build

    build.build_native()
    from pailliercryptolib_python_amd import _native

    return _native


def _is_prime(n: int) -> bool:          # independent check: Miller-Rabin on CPython ints, 12 fixed + random bases
    if n < 4 or n % 2 == 0:
        return n in (2, 3)
    d, r = n - 1, 0
    while d % 2 == 0:
        d, r = d // 2, r + 1
    for a in [2, 3, 5, 7, 11, 13] + [secrets.randbelow(n - 3) + 2 for _ in range(6)]:
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


@pytest.mark.parametrize("bits", [128, 256, 1024, 2048])
@pytest.mark.parametrize("djn", [True, False])
def test_keygen_primes_and_constraints(native, bits, djn):
    p, q = native.keygen(bits, djn)
    assert p != q and p.bit_length() == q.bit_length() == bits // 2
    assert (p * q).bit_length() == bits                       # the two top bits of each prime are set
    assert _is_prime(p) and _is_prime(q)
    assert math.gcd(p * q, (p - 1) * (q - 1)) == 1            # g = n + 1 has order n
    if djn:                                                   # upstream's DJN constraints
        assert p % 4 == 3 and q % 4 == 3 and math.gcd(p - 1, q - 1) == 2


def test_keygen_seed_is_reproducible_and_unseeded_is_not(native):
    assert native.keygen(512, True, seed=7) == native.keygen(512, True, seed=7)
    assert native.keygen(512, True, seed=7) != native.keygen(512, True, seed=8)
    assert native.keygen(512, True) != native.keygen(512, True)


def test_keygen_rejects_bad_sizes(native):
    for bits in (0, 64, 100, 130, 8256):
        with pytest.raises(native.NativeError):
            native.keygen(bits, True)


def test_host_modexp_against_cpython(native):
    rng = secrets.SystemRandom()
    for mbits in (33, 64, 65, 127, 1024, 2049, 4096, 8192):
        for _ in range(3):
            m = secrets.randbits(mbits) | 1 | (1 << (mbits - 1))
            b = secrets.randbelow(m)
            e = secrets.randbits(rng.choice([1, 5, 64, 257, 600]))
            assert native.host_modexp(b, e, m) == pow(b, e, m)
    m = (1 << 521) - 1
    assert native.host_modexp(3, 0, m) == 1 and native.host_modexp(0, 5, m) == 0 and native.host_modexp(m - 1, 2, m) == 1
    with pytest.raises(native.NativeError):
        native.host_modexp(3, 5, 1 << 64)                      # even modulus


def test_generate_keypair_uses_native_search_and_round_trips_on_ints():
    """ipclKeypair.generate_keypair builds key objects without touching a device (handles are lazy); the Paillier
    identities are checked on CPython integers with the oracle."""
    from oracle import paillier_oracle as orc
    from pailliercryptolib_python_amd.bindings import ipclKeypair

    pk, sk = ipclKeypair.generate_keypair(1024, True)
    n, hs = int(pk._n), int(pk._hs)
    p, q = int(sk._p), int(sk._q)
    assert n == p * q and n.bit_length() == 1024 and pk._randbits == 512
    key = orc.make_key(p, q, djn_x=None, bits=1024)
    m, r = 123456789, secrets.randbits(512)
    ct = (1 + m * n) * pow(hs, r, n * n) % (n * n)
    assert orc.decrypt_crt(key, ct) == m
