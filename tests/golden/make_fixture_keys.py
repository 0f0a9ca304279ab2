"""Generates tests/golden/fixture_keys.json: deterministic prime pairs for the BASELINE key sizes that
have no constants in the reference (1024/3072/4096-bit keys), plus the 2048-bit pair, which is the reference's
own bench constant pair (bench/bench_ipcl_python.py:83-97: the only fixed key material in its tree).  Primes come from the oracle's seeded
Miller-Rabin search (seed = key bits, key bits + 1), p = q = 3 mod 4.

    python tests/golden/make_fixture_keys.py
"""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from oracle import paillier_oracle as orc  # noqa: E402

out = {}
for bits in (1024, 3072, 4096):
    p = orc.seeded_prime(bits // 2, seed=bits)
    q = orc.seeded_prime(bits // 2, seed=bits + 1)
    assert p != q and (p * q).bit_length() == bits
    out[str(bits)] = {"p": hex(p), "q": hex(q)}
    print(bits, "ok", file=sys.stderr)
out["2048"] = {"p": hex(orc.BENCH_P), "q": hex(orc.BENCH_Q), "source": "reference bench/bench_ipcl_python.py:83-97"}
(Path(__file__).parent / "fixture_keys.json").write_text(json.dumps(out, indent=1) + "\n")
