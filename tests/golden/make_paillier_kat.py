"""Generates tests/golden/paillier_kat.json: known-answer vectors of the Paillier hot path, computed with nothing but
CPython integers (``pow``) from the definitions in SURVEY.md App. D — NOT through oracle/paillier_oracle.py, so that
they pin the oracle as well as the device code.  The reference pins no ciphertext bits (its tests use random keys
and assertAlmostEqual), hence "parity unpinned" for these; what they freeze is the mathematical definition on the
reference's only fixed key constants (bench/bench_ipcl_python.py:83-97) and on seeded keys of the other sizes.

    python tests/golden/make_paillier_kat.py
"""
import json
import random
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from oracle.paillier_oracle import BENCH_P, BENCH_Q  # noqa: E402  (constants only)

fx = json.loads((Path(__file__).parent / "fixture_keys.json").read_text())
keys = {2048: (BENCH_P, BENCH_Q)}
for b in (1024, 3072, 4096):
    keys[b] = (int(fx[str(b)]["p"], 16), int(fx[str(b)]["q"], 16))


def L(u, d):
    return (u - 1) // d


out = {}
for bits, (p, q) in keys.items():
    rnd = random.Random(1000 + bits)
    p, q = min(p, q), max(p, q)
    n = p * q
    nsq = n * n
    x = (1 << 70) + bits
    hs = pow((-x * x) % nsq, n, nsq)                          # DJN: hs = (-x^2)^n mod n^2
    randbits = bits // 2
    cnt = 6 if bits <= 2048 else 3
    rec = {"p": hex(p), "q": hex(q), "djn_x": hex(x), "hs": hex(hs), "randbits": randbits, "cases": []}
    lam = (p - 1) * (q - 1) // __import__("math").gcd(p - 1, q - 1)
    mu = pow(L(pow(n + 1, lam, nsq), n), -1, n)
    for i in range(cnt):
        m = [0, 1, n - 1][i] if i < 3 else rnd.randrange(n)
        r_djn = [0, (1 << randbits) - 1][i] if i < 2 else rnd.getrandbits(randbits)
        r_std = rnd.randrange(1, n)
        raw = (1 + m * n) % nsq
        ct_djn = raw * pow(hs, r_djn, nsq) % nsq
        ct_std = raw * pow(r_std, n, nsq) % nsq
        # decryption by the textbook (non-CRT) formula
        assert L(pow(ct_djn, lam, nsq), n) * mu % n == m and L(pow(ct_std, lam, nsq), n) * mu % n == m
        other = rnd.randrange(1, nsq)
        e53 = rnd.getrandbits(53) | 1
        rec["cases"].append({
            "m": hex(m), "r_djn": hex(r_djn), "r_std": hex(r_std), "raw": hex(raw), "ct_djn": hex(ct_djn), "ct_std": hex(ct_std),
            "other": hex(other), "add": hex(ct_djn * other % nsq), "e": hex(e53), "mul": hex(pow(ct_djn, e53, nsq)),
            "mul_full": hex(pow(ct_djn, n - 1 - i, nsq)), "inv": hex(pow(ct_djn, -1, nsq)),
            "pow2_7": hex(pow(ct_djn, 1 << 7, nsq)),
        })
    out[str(bits)] = rec
(Path(__file__).parent / "paillier_kat.json").write_text(json.dumps(out, indent=1) + "\n")
print({k: len(v["cases"]) for k, v in out.items()})
