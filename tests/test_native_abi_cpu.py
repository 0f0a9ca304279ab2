"""CPU: the C-ABI shared library builds, loads, exports every symbol the header declares, and fails
loudly (no CPU fallback) when asked to compute without a GPU.  No kernels are launched here."""
import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def native():
    from pailliercryptolib_python_amd import build

    build.build_native()
    from pailliercryptolib_python_amd import _native

    return _native


def test_every_header_symbol_is_exported_and_bound(native):
    header = (ROOT / "include" / "paillier_hip.h").read_text()
    declared = set(re.findall(r"\b(pai_[a-z0-9_]+)\s*\(", header))
    declared -= {"pai_pubkey", "pai_privkey", "pai_modulus"}
    lib = native.load()
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/paillier_hip.h but not exported"
    assert declared == set(native.PROTOTYPES), "ctypes prototype table and header disagree"
    assert lib.pai_version() >= 100


def test_no_cpu_fallback_without_a_device(native, gpu_available):
    if gpu_available:
        pytest.skip("a GPU is present; the no-device path is exercised on the CPU-only builder")
    lib = native.load()
    assert native.device_count() == 0
    h = C.c_void_p()
    n = np.array([0xFFFFFFFB, 0xFFFFFFFF, 0xFFFFFFFF, 0x7FFFFFFF], dtype=np.uint32)
    rc = lib.pai_pubkey_create(n.ctypes.data_as(C.c_void_p), 4, 128, None, 0, 0, 0, C.byref(h))
    assert rc == native.PAI_E_NODEVICE
    with pytest.raises(native.NativeError, match="no CPU fallback"):
        native.check(rc)
    from pailliercryptolib_python_amd import engine

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        engine.PublicKeyHandle(int.from_bytes(n.tobytes(), "little"), 128, None, 0)


def test_missing_library_is_an_import_error(native, monkeypatch, tmp_path):
    monkeypatch.setattr(native, "_lib", None)
    monkeypatch.setattr(native, "LIB_PATH", tmp_path / "libpaillier_hip.so")
    with pytest.raises(ImportError, match="no CPU fallback"):
        native.load()


def test_host_side_under_asan_ubsan():
    """sanitizer pass over the native host code (key generation, host big integers, shard plans, failure
    paths).  Builds the ASan + UBSan variant of csrc/paillier_capi.hip (host pass only, ~15 s) and runs the host-only
    entry points on it in a child process with the sanitizer runtime preloaded.  PAI_SKIP_ASAN=1 skips it."""
    import glob
    import os
    import subprocess
    import sys

    if os.environ.get("PAI_SKIP_ASAN") == "1":
        pytest.skip("PAI_SKIP_ASAN=1")
    if not glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"):
        pytest.skip("no clang ASan runtime in this image")
    from pailliercryptolib_python_amd import build

    build.build_native()
    res = subprocess.run(["bash", str(ROOT / "tools" / "asan_host_build.sh")], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    res = subprocess.run([sys.executable, str(ROOT / "tools" / "asan_host_run.py")], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "asan host run: ok" in res.stdout, (res.stdout + res.stderr)[-3000:]
