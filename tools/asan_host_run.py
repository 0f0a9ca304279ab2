"""Runs the host-only entry points of the C ABI (key generation, host modexp, shard plans, the no-device failure paths) on
the ASan/UBSan build of tools/asan_host_build.sh in a child process with the sanitizer runtime preloaded; exits non-zero on
any sanitizer report.  python tools/asan_host_run.py"""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = os.path.join(ROOT, "pailliercryptolib_python_amd", "lib", "alt", "lib_asan.so")
rt = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))
if not os.path.exists(lib) or not rt:
    sys.exit("build the library first: bash tools/asan_host_build.sh (and the clang ASan runtime must exist)")
child = r'''
import ctypes as C, secrets, numpy as np
from pailliercryptolib_python_amd import _native, engine
lib = _native.load()
for bits in (128, 512, 1024, 2048):
    for djn in (True, False):
        p, q = _native.keygen(bits, djn)
        assert (p * q).bit_length() == bits
assert _native.keygen(256, True, seed=3) == _native.keygen(256, True, seed=3)
for bits in (0, 64, 100, 8256):
    try: _native.keygen(bits, True); raise SystemExit("accepted bad size")
    except _native.NativeError: pass
for mb in (33, 64, 65, 1024, 4096, 8192):
    m = secrets.randbits(mb) | 1 | (1 << (mb - 1)); b = secrets.randbelow(m); e = secrets.randbits(200)
    assert _native.host_modexp(b, e, m) == pow(b, e, m)
try: _native.host_modexp(3, 5, 1 << 70); raise SystemExit("accepted an even modulus")
except _native.NativeError: pass
for n in (0, 1, 7, 1000, (1 << 20) + 3):
    for w in (1, 2, 3, 8):
        plan = engine.shard_plan(n, w); assert sum(c for _, c in plan) == n
h = C.c_void_p()
nn = np.array([0xFFFFFFFB, 0xFFFFFFFF, 0xFFFFFFFF, 0x7FFFFFFF], dtype=np.uint32)
rc = lib.pai_pubkey_create(nn.ctypes.data_as(C.c_void_p), 4, 128, None, 0, 0, 0, C.byref(h))
assert rc == _native.PAI_E_NODEVICE or rc == 0
print("asan host run: ok")
'''
env = dict(os.environ, PAI_NATIVE_LIB=lib, LD_PRELOAD=rt[-1], ASAN_OPTIONS="detect_leaks=0:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
           PYTHONPATH=ROOT)
res = subprocess.run([sys.executable, "-c", child], env=env, capture_output=True, text=True, cwd=ROOT)
sys.stdout.write(res.stdout[-2000:]); sys.stderr.write(res.stderr[-6000:])
bad = res.returncode != 0 or "ERROR: AddressSanitizer" in res.stderr or "runtime error:" in res.stderr
sys.exit(1 if bad else 0)
