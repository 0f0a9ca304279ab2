"""ctypes binding of ``libpaillier_hip.so`` (C ABI: ``include/paillier_hip.h``).

This module takes the place of ``from .bindings.ipcl_bindings import ...`` in the reference
(``src/ipcl_python/ipcl_python.py:5-12``).  The library is loaded from ``pailliercryptolib_python_amd/lib``
(or from the explicit path in ``PAI_NATIVE_LIB``: a variant build of the same sources); if it is missing the import of any hot-path entry fails loudly — there is no CPU
fallback behind this boundary.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

LIB_PATH = Path(__file__).resolve().parent / "lib" / "libpaillier_hip.so"
if os.environ.get("PAI_NATIVE_LIB"):     # an explicitly built variant of the same library (A/B timing of compiler flags)
    LIB_PATH = Path(os.environ["PAI_NATIVE_LIB"])

PAI_OK = 0
PAI_E_INVALID, PAI_E_NODEVICE, PAI_E_HIP, PAI_E_UNSUPPORTED, PAI_E_INTERNAL = -1, -2, -3, -4, -5


class NativeError(RuntimeError):
    """A C-ABI call failed (the reference raises RuntimeError for native failures too:
    pybind11 maps std::runtime_error, bindings/ipcl_bindings_classes.cpp:206,212,223)."""

    def __init__(self, code: int, msg: str):
        super().__init__(f"libpaillier_hip error {code}: {msg}")
        self.code = code


u32p = C.POINTER(C.c_uint32)
i32p = C.POINTER(C.c_int32)
voidp = C.c_void_p

# name -> (restype, argtypes); every symbol declared in include/paillier_hip.h
PROTOTYPES = {
    "pai_version": (C.c_int, []),
    "pai_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "pai_last_error": (C.c_char_p, []),
    "pai_profile_enable": (C.c_int, [C.c_int]),
    "pai_profile_last": (C.c_int, [C.c_int, C.c_char_p, C.c_size_t, C.POINTER(C.c_float)]),
    "pai_malloc": (C.c_int, [C.c_int, C.c_size_t, C.POINTER(voidp)]),
    "pai_free": (C.c_int, [C.c_int, voidp]),
    "pai_memcpy_h2d": (C.c_int, [C.c_int, voidp, voidp, C.c_size_t, voidp]),
    "pai_memcpy_d2h": (C.c_int, [C.c_int, voidp, voidp, C.c_size_t, voidp]),
    "pai_buf_slice": (C.c_int, [C.c_int, voidp, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, voidp, voidp]),
    "pai_buf_rotate": (C.c_int, [C.c_int, voidp, C.c_int, C.c_size_t, C.c_longlong, voidp, voidp]),
    "pai_stream_sync": (C.c_int, [C.c_int, voidp]),
    "pai_host_stage": (C.c_int, [C.c_int, C.c_int, C.POINTER(voidp), C.POINTER(C.c_size_t), voidp, C.POINTER(voidp)]),
    "pai_ct_add_aligned_host": (C.c_int, [voidp, voidp, voidp, C.c_int, voidp, C.c_size_t, voidp, voidp]),
    "pai_ct_mul_host": (C.c_int, [voidp, voidp, voidp, C.c_int, C.c_int, C.c_int, C.c_size_t, voidp, voidp]),
    "pai_pubkey_create": (C.c_int, [voidp, C.c_int, C.c_int, voidp, C.c_int, C.c_int, C.c_int, C.POINTER(voidp)]),
    "pai_pubkey_destroy": (None, [voidp]),
    "pai_keygen": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_uint64), voidp, voidp]),
    "pai_host_modexp": (C.c_int, [voidp, voidp, C.c_int, voidp, C.c_int, voidp]),
    "pai_pubkey_info": (C.c_int, [voidp] + [C.POINTER(C.c_int)] * 7),
    "pai_pubkey_table_info": (C.c_int, [voidp, C.POINTER(C.c_size_t), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "pai_path_edges": (C.c_int, [voidp, C.c_int, C.POINTER(C.c_size_t), C.c_int, C.POINTER(C.c_int)]),
    "pai_privkey_create": (C.c_int, [voidp, voidp, C.c_int, voidp, C.c_int, C.POINTER(voidp)]),
    "pai_privkey_destroy": (None, [voidp]),
    "pai_raw_encrypt": (C.c_int, [voidp, voidp, C.c_size_t, voidp, voidp]),
    "pai_ct_add_plain": (C.c_int, [voidp, voidp, voidp, C.c_size_t, voidp, voidp]),
    "pai_encrypt": (C.c_int, [voidp, voidp, voidp, C.c_size_t, voidp, voidp]),
    "pai_obfuscate": (C.c_int, [voidp, voidp, voidp, C.c_size_t, voidp]),
    "pai_decrypt": (C.c_int, [voidp, voidp, C.c_size_t, voidp, voidp]),
    "pai_ct_add": (C.c_int, [voidp, voidp, voidp, C.c_int, C.c_size_t, voidp, voidp]),
    "pai_ct_mul": (C.c_int, [voidp, voidp, voidp, C.c_int, C.c_int, C.c_int, C.c_size_t, voidp, voidp]),
    "pai_ct_invert": (C.c_int, [voidp, voidp, C.c_size_t, voidp, voidp]),
    "pai_ct_invert_async": (C.c_int, [voidp, voidp, C.c_size_t, voidp, voidp]),
    "pai_ct_invert_flag": (C.c_int, [voidp, voidp, C.c_size_t, voidp, voidp, voidp]),
    "pai_pubkey_status": (C.c_int, [voidp, C.POINTER(C.c_int), C.c_int, voidp]),
    "pai_ct_add_aligned": (C.c_int, [voidp, voidp, voidp, C.c_int, voidp, C.c_size_t, voidp, voidp]),
    "pai_ct_add_aligned_dom": (C.c_int, [voidp, voidp, voidp, C.c_int, voidp, C.c_size_t, voidp, voidp, voidp]),
    "pai_ct_mont_mul": (C.c_int, [voidp, voidp, voidp, C.c_int, C.c_size_t, voidp, voidp]),
    "pai_ct_addn": (C.c_int, [voidp, C.POINTER(voidp), C.POINTER(voidp), C.c_int, C.c_int, C.c_int, C.c_int, C.c_size_t, voidp, voidp]),
    "pai_pubkey_mont_bits": (C.c_int, [voidp, C.POINTER(C.c_int)]),
    "pai_ct_prod": (C.c_int, [voidp, voidp, C.c_size_t, C.c_size_t, voidp, voidp]),
    "pai_ct_multiexp": (C.c_int, [voidp, voidp, voidp, C.c_size_t, C.c_size_t, C.c_size_t, voidp, C.c_int, C.c_int, voidp, voidp,
                                  voidp]),
    "pai_shard_plan": (C.c_int, [C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "pai_gather": (C.c_int, [C.c_int, i32p, C.POINTER(voidp), C.POINTER(C.c_size_t), C.c_int, C.c_int, voidp]),
    "pai_scatter": (C.c_int, [C.c_int, i32p, C.POINTER(voidp), C.POINTER(C.c_size_t), C.c_int, C.c_int, voidp]),
    "pai_ct_pow2": (C.c_int, [voidp, voidp, voidp, C.c_int, C.c_size_t, voidp]),
    "pai_pubkey_trim": (C.c_int, [voidp, C.POINTER(C.c_size_t)]),
    "pai_ct_pow2_hint": (C.c_int, [voidp, voidp, voidp, C.c_int, C.c_size_t, C.c_int, voidp]),
    "pai_fp_encode_f64": (C.c_int, [voidp, voidp, C.c_size_t, voidp, voidp, voidp]),
    "pai_fp_encode_i64": (C.c_int, [voidp, voidp, C.c_size_t, voidp, voidp, voidp]),
    "pai_fp_encode_at": (C.c_int, [voidp, voidp, C.c_int, C.c_size_t, voidp, C.c_int, voidp, voidp, voidp]),
    "pai_fp_decode_i64": (C.c_int, [voidp, voidp, C.c_size_t, voidp, voidp, voidp]),
    "pai_draw_r": (C.c_int, [voidp, voidp, voidp, C.c_uint32, C.c_size_t, voidp, voidp]),
    "pai_modulus_create": (C.c_int, [voidp, C.c_int, C.c_int, C.POINTER(voidp)]),
    "pai_modulus_destroy": (None, [voidp]),
    "pai_modmul": (C.c_int, [voidp, voidp, voidp, C.c_int, C.c_size_t, voidp, voidp]),
    "pai_modexp_fixed": (C.c_int, [voidp, voidp, voidp, C.c_int, C.c_size_t, voidp, voidp]),
    "pai_modexp_var": (C.c_int, [voidp, voidp, C.c_int, voidp, C.c_int, C.c_int, C.c_int, C.c_size_t, voidp, voidp]),
}

_lib = None
# host-side limits of the native key helpers (csrc/host_keygen.hpp: kg::MAXL 64-bit limbs): pai_keygen serves keys up to
# KEYGEN_MAX_BITS, pai_host_modexp odd moduli up to HOST_MODEXP_MAX_WORDS 32-bit words; callers fall back to CPython ints above
KEYGEN_MAX_BITS = 8192
HOST_MODEXP_MAX_WORDS = 260


def load() -> C.CDLL:
    """dlopen the library (once) and attach prototypes.  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    build_error = None
    try:
        from . import build as _build

        if _build.LIB == LIB_PATH:
            if not LIB_PATH.exists():
                # not built yet (fresh checkout): compile it in-tree with hipcc; a build step, not a fallback
                _build.build_native()
            elif LIB_PATH.stat().st_mtime < _build._deps_mtime():
                # never rebuilt implicitly (several ranks may be importing at once): say so instead
                import warnings

                warnings.warn("libpaillier_hip.so is older than csrc/: run `python -m pailliercryptolib_python_amd.build`")
    except Exception as e:      # noqa: BLE001 - reported below if the library is unusable
        build_error = e
    if not LIB_PATH.exists():
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m pailliercryptolib_python_amd.build` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback."
            + (f"  The in-tree build failed: {build_error}" if build_error else "")
        ) from build_error
    lib = C.CDLL(str(LIB_PATH))
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)          # AttributeError here = ABI/header mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(code: int) -> None:
    if code != PAI_OK:
        msg = load().pai_last_error()
        raise NativeError(code, msg.decode("utf-8", "replace") if msg else "")


def keygen(key_bits: int, djn: bool, seed=None):
    """pai_keygen: two random primes (p, q) for a key of key_bits bits as Python ints (host-only native search)."""
    import numpy as np

    words = key_bits // 64
    p = np.zeros(words, dtype=np.uint32)
    q = np.zeros(words, dtype=np.uint32)
    sd = C.byref(C.c_uint64(int(seed) & (2**64 - 1))) if seed is not None else None
    check(load().pai_keygen(int(key_bits), 1 if djn else 0, sd, p.ctypes.data, q.ctypes.data))
    return int.from_bytes(p.tobytes(), "little"), int.from_bytes(q.tobytes(), "little")


def host_modexp(base: int, exp: int, mod: int) -> int:
    """pai_host_modexp: base^exp mod an odd modulus on the host cores (key set-up; not a hot operation)."""
    import numpy as np

    mw = (mod.bit_length() + 31) // 32
    ew = max(1, (exp.bit_length() + 31) // 32)
    b = np.frombuffer((base % mod).to_bytes(4 * mw, "little"), dtype=np.uint32).copy()
    e = np.frombuffer(exp.to_bytes(4 * ew, "little"), dtype=np.uint32).copy()
    m = np.frombuffer(mod.to_bytes(4 * mw, "little"), dtype=np.uint32).copy()
    out = np.zeros(mw, dtype=np.uint32)
    check(load().pai_host_modexp(b.ctypes.data, e.ctypes.data, ew, m.ctypes.data, mw, out.ctypes.data))
    return int.from_bytes(out.tobytes(), "little")


def device_count() -> int:
    n = C.c_int(0)
    check(load().pai_device_count(C.byref(n)))
    return n.value
