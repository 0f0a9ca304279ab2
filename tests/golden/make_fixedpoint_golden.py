"""Generates tests/golden/fixedpoint_golden.json by running the REFERENCE's own codec
(/root/reference/src/ipcl_python/bindings/fixedpoint.py, imported standalone in the build container;
it cannot travel to the GPU box) on a fixed list of scalars.  Only inputs and outputs are stored.

Each record: {"t": type tag, "v": value (float.hex() / decimal int string), "enc": hex encoding,
"exp": exponent, "dec": decoded value (float.hex() or "int:<decimal>")}  or  {"err": exception class}
for inputs the reference rejects; "dec_err" when decode (of a hand-made encoding) raises.

    python tests/golden/make_fixedpoint_golden.py
"""
import importlib.util
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from oracle.paillier_oracle import BENCH_P, BENCH_Q  # noqa: E402

spec = importlib.util.spec_from_file_location("ref_fixedpoint", "/root/reference/src/ipcl_python/bindings/fixedpoint.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)
FP = ref.FixedPointNumber

N_KEY = BENCH_P * BENCH_Q
MAX_INT = N_KEY // 3 - 1

TAGS = {
    "float": float, "int": int, "bool": bool,
    "np.float64": np.float64, "np.float32": np.float32, "np.int64": np.int64, "np.int32": np.int32,
    "np.int16": np.int16, "np.int8": np.int8, "np.uint8": np.uint8,
}


def tag_of(v):
    for k, t in TAGS.items():
        if type(v) is t:
            return k
    raise TypeError(type(v))


def ser(v):
    if isinstance(v, (float, np.floating)):
        return float(v).hex()
    return str(int(v))


def dec_ser(d):
    return "int:%d" % d if isinstance(d, int) else float(d).hex()


scalars = [
    0.0, -0.0, 5e-324, 2.2e-308, 9.99e-201, 1e-200, -1e-200, 1.0, -1.0, 0.5, -0.2, 0.1, 1234.5678, -5111.2834,
    2.0**52, 2.0**53, 2.0**53 + 2, 2.0**60, -(2.0**70), 1e300, -1e300, sys.float_info.max, float("inf"), float("-inf"),
    float("nan"), 3.141592653589793, 1 / 3, -1 / 3, 1e-5, 123456789.125,
    0, 1, -1, 2, 255, 256, 2**32 - 1, 2**32, -(2**32), 2**63, 10**30, -(10**30), MAX_INT, -MAX_INT, MAX_INT + 1,
    1 << 2046, True, False, -(2**63), -(2**63) + 1, -(2**63) - 1, np.int64(-(2**63)),
    np.float64(2.5), np.float64(-1e-3), np.float32(1.5), np.int64(-77), np.int32(12345), np.int16(-3), np.int8(5),
    np.uint8(5),
]
rng = np.random.default_rng(20240929)
scalars += [float(v) for v in rng.uniform(-1000, 1000, 60)]
scalars += [float(v) for v in rng.normal(0, 1, 30) * 10.0 ** rng.integers(-30, 30, 30)]
scalars += [int(v) for v in rng.integers(-(2**62), 2**62, 20)]

records = []
for v in scalars:
    rec = {"t": tag_of(v), "v": ser(v) if not (isinstance(v, float) and v != v) else "nan"}
    try:
        with np.errstate(all="ignore"):
            e = FP.encode(v, N_KEY, MAX_INT)
        rec["enc"] = hex(e.encoding)
        rec["exp"] = int(e.exponent)
        rec["dec"] = dec_ser(e.decode())
    except Exception as ex:  # noqa: BLE001
        rec["err"] = type(ex).__name__
    records.append(rec)

# decode-only cases on hand-made encodings
dec_cases = []
for enc, expo in [(0, 0), (1, 0), (MAX_INT, 0), (MAX_INT + 1, 0), (N_KEY // 2, 3), (N_KEY - MAX_INT, 0), (N_KEY - MAX_INT - 1, 0),
                  (N_KEY - 1, 0), (N_KEY - 1, 10), (N_KEY, 0), (N_KEY + 5, 0), (12345, 7), (12345, -7), (N_KEY - 12345, 52),
                  ((1 << 200) + 1, 150), (N_KEY - (1 << 200) - 1, 150), (3 << 60, 61)]:
    rec = {"enc": hex(enc), "exp": expo}
    try:
        rec["dec"] = dec_ser(FP(enc, expo, N_KEY, MAX_INT).decode())
    except Exception as ex:  # noqa: BLE001
        rec["dec_err"] = type(ex).__name__
    dec_cases.append(rec)

out = {"n": hex(N_KEY), "max_int": hex(MAX_INT), "encode": records, "decode": dec_cases,
       "source": "src/ipcl_python/bindings/fixedpoint.py (reference, imported standalone)", "numpy": np.__version__}
(Path(__file__).parent / "fixedpoint_golden.json").write_text(json.dumps(out, indent=0) + "\n")
print(len(records), "encode records,", len(dec_cases), "decode records;",
      sum("err" in r for r in records), "encode errors")
