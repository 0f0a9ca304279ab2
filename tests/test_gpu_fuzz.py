"""GPU: a bounded, seeded pass of the differential fuzz (tools/fuzz_gpu.py) inside the driver's suite.

The fuzz is the one thing that crosses every path switch of the C ABI at random batch sizes, key sizes (1024 ... 4096 bits, with
sizes between the geometries) and operand patterns (carry chains, values next to 0 and M), every result against CPython integers.
Rounds 2-5 ran it by hand (profiles/r0N/fuzz_*.json); VERDICT r05 asked for it where the driver sees it."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


@pytest.mark.parametrize("seed", [20260930])
def test_differential_fuzz_bounded(seed):
    env = dict(os.environ)
    for k in ("PAI_LATENCY_MAX", "PAI_TUNE", "PAI_DISABLE", "PAI_LAT_ADD_MAX", "PAI_POW2_DIGIT_MIN"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, str(ROOT / "tools" / "fuzz_gpu.py"), "15", str(seed)], capture_output=True, text=True,
                         cwd=str(ROOT), env=env, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    line = json.loads(res.stdout.strip().splitlines()[-1])
    assert line["failures"] == 0 and line["rounds"] >= 3 and line["seed"] == seed, line
