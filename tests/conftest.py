"""pytest configuration: registers the `gpu` marker and makes the repo root importable.

`-m "not gpu"` runs here (no GPU): oracle vs golden vectors, host logic, C-ABI symbol checks.
`-m gpu` runs on an MI355X: parity of the HIP path against the oracle through the C ABI.
"""
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950) device")


def _has_gpu() -> bool:
    try:
        from pailliercryptolib_python_amd import _native

        return _native.device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu_available():
    return _has_gpu()
