"""Builds the native HIP library (libpaillier_hip.so) for gfx950 in-tree.

Replaces the reference's CMake/FetchContent build (CMakeLists.txt, lib/ipcl.cmake, setup.py:38-60):
there is nothing to fetch — every kernel is in ``csrc/`` — and the only tool needed is ``hipcc``.
The objects are compiled in parallel, one translation unit per lane-group geometry.

    python -m pailliercryptolib_python_amd.build [--force]
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
OBJ = CSRC / "build"
LIBDIR = PKG / "lib"
LIB = LIBDIR / "libpaillier_hip.so"
ARCH = "gfx950"

SOURCES = [
    "geo_36x1.hip",
    "geo_36x2.hip",
    "geo_28x4.hip",
    "geo_36x4.hip",
    "geo_28x8.hip",
    "geo_36x8.hip",
    "geo_3x16.hip",
    "geo_3x32.hip",
    "geo_3x64.hip",
    "geo_9x32.hip",
    "inv_eea.hip",
    "wide_kernels.hip",
    "padic_dec_kernels.hip",
    "padic_enc_kernels.hip",
    "padic_enc36_kernels.hip",
    "pair_kernels.hip",
    "paillier_capi.hip",
]

# Per-translation-unit extra flags.  The digit-pair kernels run one wave per SIMD, so their speed is set by how the
# compiler orders ~6000 instructions per product.  Scheduler strategies were A/B-timed on one MI355X box with
# tools/build_variants.sh: -amdgpu-sched-strategy=max-ilp wins 5-11 % in the isolated probes of tools/padic_bench.hip
# but LOSES 4 % inside k_dec_a_padic (508 vs 487 ms); -amdgpu-use-amdgpu-trackers and -amdgpu-schedule-metric-bias=0
# are within noise (483-485 ms).  Hence: none.
EXTRA_FLAGS: dict = {}


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _deps_mtime() -> float:
    files = list(CSRC.glob("*.hip")) + list(CSRC.glob("*.hpp")) + [PKG.parent / "include" / "paillier_hip.h"]
    return max(f.stat().st_mtime for f in files)


def _compile(src: str) -> None:
    obj = OBJ / (Path(src).stem + ".o")
    if obj.exists() and obj.stat().st_mtime >= _deps_mtime():
        return
    # the row engines rely on full unrolling of loops with thousands of multiply-accumulates (register-resident
    # accumulator windows need compile-time indices); clang's default pragma-unroll budget silently falls back to a
    # partial unroll there, which demotes the window to scratch memory (4x slower)
    cmd = [_hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-mllvm", "-pragma-unroll-threshold=1048576",
           *EXTRA_FLAGS.get(src, []), "-c", str(CSRC / src), "-o", str(obj)]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{res.stderr[-4000:]}")


def build_native(force: bool = False, jobs: int | None = None, verbose: bool = False) -> Path:
    """Compile (if stale) and return the path of libpaillier_hip.so."""
    OBJ.mkdir(parents=True, exist_ok=True)
    LIBDIR.mkdir(parents=True, exist_ok=True)
    if force:
        for o in OBJ.glob("*.o"):
            o.unlink()
    if LIB.exists() and not force and LIB.stat().st_mtime >= _deps_mtime():
        return LIB
    jobs = jobs or min(len(SOURCES), max(1, (os.cpu_count() or 2) - 1))
    if verbose:
        print(f"[build] hipcc --offload-arch={ARCH}: {len(SOURCES)} translation units, {jobs} jobs", flush=True)
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        list(ex.map(_compile, SOURCES))
    objs = [str(OBJ / (Path(s).stem + ".o")) for s in SOURCES]
    cmd = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", str(LIB)] + objs
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"link failed:\n{res.stderr[-4000:]}")
    return LIB


if __name__ == "__main__":
    path = build_native(force="--force" in sys.argv, verbose=True)
    print(path)
