#!/usr/bin/env python3
"""Headline benchmark: Paillier encrypt+decrypt ops/sec, 2048-bit key, batch = 1 M.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config headline|cfg2|cfg4|cfg5] [--key-bits B] [--batch B]
                    [--scaling strong|weak] [--no-cpu-baseline] [--no-extras] [--no-reference-bench]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`python bench.py --gpus N` with N > 1 and no torchrun environment launches the N ranks itself (one process per GPU under
torch.distributed.run, rendezvous on 127.0.0.1) and fails loudly when fewer than N GPUs are visible.

One "step" = one pass of the hot path over one synthetic batch that is already resident in HBM:
DJN-obfuscated encryption of the plaintext residues (pai_encrypt) followed by CRT decryption of the
ciphertexts (pai_decrypt).  One op = one element encrypted AND decrypted (BASELINE.json metric,
SURVEY.md §8d).  The path shards by independent elements (no data-path collective):
  --scaling strong (default) BASELINE's "batch = 1 M on 1/2/4/8 GPUs": the B elements are split over the ranks by
                   the contiguous block partition of pai_shard_plan; the final RCCL gather of the ciphertext shards
                   is timed separately and reported as gather_ms (it is not part of an op).  With N > 1 the line
                   also carries the other arrangement under "weak_scaling" (same K steps, timed the same way);
  --scaling weak   every rank runs B elements on its own device ("strong_scaling" then carries the other one).
The timed region is bracketed by a barrier + torch.cuda.synchronize() and the maximum over ranks is taken.

Inputs: key = the reference's bench constants P, Q (bench/bench_ipcl_python.py:83-97, stored in
tests/golden/fixture_keys.json) with a fixed DJN base — built here from plain integers, the oracle only checks;
plaintexts = fixed-point encodings of default_rng(1002).uniform(-1000, 1000, B); randomness r = seeded device
generator (1024 random bits per element), all uploaded before the timed region.

The JSON line also carries
  roofline     — the dominant kernel (k_dec_a: the two CRT half-size modexps) against the measured v_mad_u64_u32
                 peak of the chip (profiles/r01/ubench_valu_mi355x.jsonl); the path is integer-VALU bound, so the
                 "bound" is "valu_int".  frac = EXECUTED multiply-accumulates / peak (kernel quality, <= 1);
                 canonical_frac prices the kernel at the canonical algorithm's work and can
                 exceed 1 because the kernel runs a cheaper algorithm.  traffic = HBM bytes per launch from the PMC
                 passes committed under profiles/ (file and sha256 named; null when no such file is present).
  cpu_baseline — the same two operations on the host cores on a bounded sample: the AVX512-IFMA mb8 port
                 (oracle/paillier_ifma.c — the algorithm of README.md:32's mbx_exp_mb8) when the host has IFMA,
                 else libgmp, else the plain-C port.
  api_level    — host float64 array -> PaillierPublicKey.encrypt -> PaillierPrivateKey.decrypt_to_numpy -> host
                 float64 array (codec, CSPRNG-keyed randomness and PCIe included): SURVEY.md §8d(i).
  other_ops    — BASELINE configs[2]: ct+ct add, ct x pt mul (and ct^-1, sum) on the same resident batch, each
                 checked against the oracle; small_batch — latency at the reference's own batch sizes 16 / 64 and at a mid-size batch (8 192).
  reference_bench — the reference's own benchmark suite (bench/bench_ipcl_python.py:13-78: KeyGen, Encrypt, Decrypt,
                 Add_CTCT, Add_CTPT, Mul_CTPT at 16 / 64 with its inputs and key) through the PUBLIC API, microseconds per
                 call, with the same composition on the CPU port beside it and a bit-for-bit parity check of every
                 deterministic row (2048-bit configurations only).
  configs      — default single-GPU invocation only: BASELINE configs[1] (2048-bit, 65 536), configs[3] (3072-bit, 2^20) and
                 configs[4] (4096-bit, 2^18) timed in the same run after the headline leg (--config-steps timed steps each),
                 each with value, ms_per_step, the full-batch parity check, per-kernel times, roofline and a CPU sample.
--config selects the BASELINE.json configuration: headline (2048-bit, 2^20: the metric), cfg2 (2048-bit, 65 536),
cfg4 (3072-bit, 2^20), cfg5 (4096-bit, 2^18): key size, batch, executed / canonical MAC counts, dominant kernel and the
PMC file for roofline.traffic follow the key size; the parity check and cpu_baseline are the same.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import time
from pathlib import Path
from types import SimpleNamespace
from typing import Optional

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
# the CPU-baseline library is OpenMP: idle workers must sleep, not spin (a spinning pool eats the cgroup CPU quota of
# the single-thread small-batch timings that follow a parallel region); read by libgomp when it is first loaded
os.environ.setdefault("OMP_WAIT_POLICY", "passive")

DJN_X = 0x1234567
# v_mad_u64_u32 rate of the chip, 8 waves/SIMD: PEAK = an 18 ms burst on constant operands (profiles/r01/ubench_valu_mi355x.jsonl);
# PEAK_SUSTAINED = seconds-long back-to-back launches on data-dependent operands with the clock the power management settles
# at (2.29 GHz / 1290 W of the 1400 W cap; profiles/r04/ubench_valu_sustained.jsonl, form 1) - the two differ by 2.4 %
PEAK_MAC32_PER_S = 35.9e12
PEAK_SUSTAINED_MAC32_PER_S = 35.05e12
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md
# canonical MAC32 per op (CIOS 2L^2+L, 5-bit window): encrypt DJN, decrypt CRT, ct+ct, ct x pt (53-bit e)
CANON = {1024: (5.35e6, 2.70e6, 16.5e3, 0.79e6), 2048: (41.52e6, 20.86e6, 65.8e3, 3.16e6),
         3072: (138.76e6, 69.61e6, 147.8e3, 7.10e6), 4096: (327.15e6, 163.99e6, 262.7e3, 12.61e6)}
BASELINE_METRIC = json.loads((ROOT / "BASELINE.json").read_text())["metric"] if (ROOT / "BASELINE.json").exists() else \
    "Paillier encrypt+decrypt ops/sec, 2048-bit key, batch=1M; 1/2/4/8 MI355X"
# BASELINE.json configurations this tool can time (same step, same parity check, same roofline / cpu_baseline objects):
# the metric's own configuration and the two sharded ones (cfg4 / cfg5 = BASELINE.json configs[3] / configs[4])
CONFIGS = {
    "headline": {"key_bits": 2048, "batch": 1 << 20, "baseline": "metric (configs[1]/[2] key and batch): 2048-bit key, batch 2^20"},
    "cfg2": {"key_bits": 2048, "batch": 1 << 16, "baseline": "configs[1]: 2048-bit key, batch 65 536"},
    "cfg4": {"key_bits": 3072, "batch": 1 << 20, "baseline": "configs[3]: 3072-bit key, batch 2^20 encrypt+decrypt sharded over the ranks"},
    "cfg5": {"key_bits": 4096, "batch": 1 << 18, "baseline": "configs[4]: 4096-bit key, batch 2^18 encrypt with CRT decrypt"},
}
# PMC summaries by key size, newest first: (file, batch the profile was taken at)
PMC_FILES = {2048: [("profiles/r06/pmc_bench_r06.json", 1 << 20), ("profiles/r05/pmc_bench_r05.json", 1 << 20), ("profiles/r04/pmc_bench_r04.json", 1 << 20), ("profiles/r03/pmc_bench_r03.json", 1 << 20),
                    ("profiles/r02/pmc_bench_r02.json", 1 << 20), ("profiles/r01/pmc_bench_r01d.json", 1 << 20)],
             3072: [("profiles/r05/pmc_cfg4_r05.json", 1 << 20), ("profiles/r04/pmc_cfg4_r04.json", 1 << 20), ("profiles/r03/pmc_k3072_r03.json", 1 << 16)],
             4096: [("profiles/r05/pmc_cfg5_r05.json", 1 << 18), ("profiles/r04/pmc_cfg5_r04.json", 1 << 18), ("profiles/r03/pmc_k4096_r03.json", 1 << 16)]}


def alg_bytes(bits: int):
    """Algorithmic bytes per op: encrypt 8 + k/16 in, k/4 out; decrypt k/4 in, 8 out; add 3 k/4."""
    return 8 + bits // 16 + bits // 4, bits // 4 + 8, 3 * bits // 4


def padic_nl(prime_bits: int) -> int:
    """Limbs of the digit engine serving a prime (csrc/padic_dec_kernels.hip: padic_nl_for_prime_bits)."""
    for nl in (24, 36, 56, 72):
        if 29 * nl >= prime_bits + 20 and prime_bits >= 400:
            return nl
    return 0


def modmul_limbs(mod_bits: int) -> int:
    """Limbs of the lane-group geometry serving a modulus (csrc/geo_inst.hpp: geo_for_bits)."""
    for nl in (36, 72, 112, 144, 224, 288):
        if 29 * nl >= mod_bits + 2:
            return nl
    return 0


def _sliding_counts(e: int, w: int = 6):
    """(#squarings, #multiplications) of the left-to-right sliding-window schedule the library compiles for the
    exponent e (csrc/paillier_capi.hip): windows of at most w bits that end in a 1."""
    i, nsq, nmul, first, pending = e.bit_length() - 1, 0, 0, True, 0
    while i >= 0:
        if not (e >> i) & 1:
            pending += 1
            i -= 1
            continue
        l = min(w, i + 1)
        while not (e >> (i - l + 1)) & 1:
            l -= 1
        if not first:
            nsq += pending + l
            nmul += 1
        first, pending = False, 0
        i -= l
    return nsq + pending, nmul


def synthetic_key(bits: int = 2048, djn_x: Optional[int] = 0x1234567) -> SimpleNamespace:
    """Synthetic key material for throughput runs: the prime pair of tests/golden/fixture_keys.json (2048 bits = the
    reference's own bench constants, bench/bench_ipcl_python.py:83-97; other sizes seeded) and, for DJN, the
    obfuscator base hs = (-x^2)^n mod n^2 with randbits = bits / 2.  Plain CPython integers."""
    fx = json.loads((Path(__file__).resolve().parent / "tests" / "golden" / "fixture_keys.json").read_text())[str(bits)]
    p, q = sorted((int(fx["p"], 16), int(fx["q"], 16)))
    n = p * q
    nsq = n * n
    hs = pow((-djn_x * djn_x) % nsq, n, nsq) if djn_x is not None else None
    return SimpleNamespace(bits=bits, p=p, q=q, n=n, nsq=nsq, hs=hs, randbits=bits // 2 if djn_x is not None else 0,
                           max_int=n // 3 - 1, djn_x=djn_x)


def executed_macs_decrypt(p: int, q: int, w: int = 6) -> float:
    """29x29-bit MACs actually issued per decrypted element by k_dec_a_padic (both primes): on base-s digit
    pairs of NL limbs a multiplication takes 5 NL^2 MACs and a squaring NL^2 (reduction of a^2) + 2 NL^2 (2ab and its
    reduction) + the a^2 half: NL (NL+1)/2 up to 36 limbs (every limb pair once), NL (NL+8)/2 at 56 limbs (limb-class
    symmetric in 8-row blocks: csrc/mont_padic.hpp sqr_sym_fused), NL^2 at 72 limbs (sqr_fused)."""
    nl = padic_nl(max(p.bit_length(), q.bit_length()))
    half = {24: nl * (nl + 1) // 2, 36: nl * (nl + 1) // 2, 56: nl * (nl + 8) // 2, 72: nl * nl}[nl]
    sq = half + 3 * nl * nl
    mul = 5 * nl * nl
    ct_bits = 2 * (p * q).bit_length()
    nd = -(-ct_bits // (29 * nl))
    total = 0.0
    for s_ in (p, q):
        n_sq, n_mul = _sliding_counts(s_ - 1, w)
        n_sq += 1                                   # base^2 for the odd-power table
        n_mul += (1 << (w - 1)) - 1 + 1             # table of odd powers + leaving Montgomery form
        total += n_sq * sq + n_mul * mul + nd * 4 * nl * nl
    return total


def executed_macs_std_obfuscator(n: int, w: int = 6) -> float:
    """29x29-bit MACs per element of the standard scheme's obfuscator r^n mod n^2 on base-n digit pairs (k_pow_padic,
    csrc/kernels_padic_enc.hpp): the sliding-window schedule of n (windows of <= 6 bits), squarings at 4 NL^2, window and
    table products (2^(w-1) odd powers) and the two entry products at 5 NL^2; plus the fused (1 + m n) r^n product."""
    nl = padic_nl(n.bit_length())
    nsq, nmul = _sliding_counts(n, w)
    return nsq * 4.0 * nl * nl + (nmul + (1 << (w - 1)) + 2 + 1) * 5.0 * nl * nl


def pmc_traffic(kernel_prefix: str, batch: int, key_bits: int = 2048):
    """HBM bytes per launch of a kernel from the newest committed PMC summary of this key size (rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE in separate passes, KiB units; gfx950 FETCH_SIZE counts half of a wide streaming read => doubled, as
    MI355X_MICROARCH.md prescribes).  Scaled linearly from the batch the profile was taken at to `batch`."""
    for rel, prof_batch in PMC_FILES.get(key_bits, []):
        f = ROOT / rel
        if not f.exists():
            continue
        raw = f.read_bytes()
        data = json.loads(raw)
        for name, c in data.items():
            if name.startswith(kernel_prefix) and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                scale = batch / float(prof_batch)
                return {"bytes": (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0 * scale, "fetch_KiB": c["FETCH_SIZE"],
                        "write_KiB": c["WRITE_SIZE"], "source": rel, "profile_batch": prof_batch,
                        "sha256": hashlib.sha256(raw).hexdigest()[:16], "kernel": name}
    return None


def reference_bench(key, okey, device) -> dict:
    """The reference's own benchmark suite (bench/bench_ipcl_python.py:13-78) through the PUBLIC API, with
    its exact inputs and its fixed 2048-bit key: BM_KeyGen 1024 / 2048 and BM_Encrypt / BM_Decrypt / BM_Add_CTCT /
    BM_Add_CTPT / BM_Mul_CTPT at 16 and 64 elements, microseconds per call (the unit the reference prints).  Every call is
    followed by a device synchronisation (the reference's calls are synchronous).  Beside each row: the same
    composition (ipcl_python.py's encode / align / raw-encrypt / modexp / modmul sequence, oracle/c_oracle.CApi) on the
    CPU port's AVX512-IFMA mb8 kernels with one thread and with all granted threads.  The deterministic rows are checked
    bit for bit: the CPU leg is fed the GPU's input ciphertexts and must return the GPU's output ciphertexts."""
    import torch

    from oracle import c_oracle as co
    from pailliercryptolib_python_amd import PaillierKeypair, PaillierPrivateKey, PaillierPublicKey, engine
    from pailliercryptolib_python_amd.bindings import ipclPublicKey

    pk = PaillierPublicKey(ipclPublicKey(key.n, key.bits, True, hs=key.hs, randbits=key.randbits, device=device))
    sk = PaillierPrivateKey(pk, key.p, key.q)
    threads = co.max_threads()
    capis = {"cpu_1_thread_us": co.CApi(okey, threads=1), f"cpu_{threads}_threads_us": co.CApi(okey, threads=threads)}

    def us(f, sync: bool, budget_s: float = 0.4, min_reps: int = 3, max_reps: int = 200):
        f()                                                      # warm (first call: tables, scratch, thread pool)
        if sync:
            torch.cuda.synchronize()
        n, t0 = 0, time.perf_counter()
        while True:
            f()
            if sync:
                torch.cuda.synchronize()
            n += 1
            el = time.perf_counter() - t0
            if n >= max_reps or (n >= min_reps and el > budget_s):
                return 1e6 * el / n

    def host(enc):
        return engine.to_host_words(enc.words), enc.exponent()

    out = {"unit": "microseconds per call (mean)", "key": "reference bench P, Q (2048 bits), DJN",
           "source": "bench/bench_ipcl_python.py:13-78", "cpu_kind": "port (IFMA mb8) driven through ipcl_python.py's composition"
           if co.ifma_available() else "port (plain C / CPython pow) driven through ipcl_python.py's composition", "rows": {}}
    for bits in (1024, 2048):
        out["rows"][f"BM_KeyGen/{bits}"] = {
            "gpu_api_us": us(lambda: PaillierKeypair.generate_keypair(bits), sync=False, budget_s=1.0),
            "note": "pai_keygen: host-side sieve + Miller-Rabin on two threads, DJN base through pai_host_modexp; no CPU-port "
                    "counterpart (the reference's generator is IPP-Crypto's, absent here)"}
    # the key-generation rows above are host-only: the device's clocks have dropped meanwhile — bring them back before the first
    # device row is timed (0.3 s of the calls the rows make)
    t_warm = time.perf_counter()
    while time.perf_counter() - t_warm < 0.3:
        sk.decrypt(pk.encrypt(np.arange(16) * 1.5))
    torch.cuda.synchronize()
    for nb in (16, 64):
        ar = np.arange(nb)
        x_enc, x_dec = (ar + 11) * 1234.5678, (ar + 1) * 1234.5678
        x, y = (ar + 11) * 5111.2834, (32768 - ar) * 1.3872
        ct_dec, ct_x, ct_y = pk.encrypt(x_dec), pk.encrypt(x), pk.encrypt(y)
        ct_xx = ct_x * x
        cases = {
            "BM_Encrypt": (lambda: pk.encrypt(x_enc), lambda c: c.encrypt(x_enc)),
            "BM_Decrypt": (lambda: sk.decrypt(ct_dec), None),
            "BM_Add_CTCT": (lambda: ct_x + ct_y, None),
            "BM_Add_CTPT": (lambda: ct_xx + y, None),
            "BM_Mul_CTPT": (lambda: ct_x * y, None),
        }
        h_dec, h_x, h_y, h_xx = host(ct_dec), host(ct_x), host(ct_y), host(ct_xx)
        cpu_f = {
            "BM_Encrypt": lambda c: c.encrypt(x_enc),
            "BM_Decrypt": lambda c: c.decrypt(*h_dec),
            "BM_Add_CTCT": lambda c: c.add_ctct(*h_x, *h_y),
            "BM_Add_CTPT": lambda c: c.add_ctpt(*h_xx, y),
            "BM_Mul_CTPT": lambda c: c.mul_ctpt(*h_x, y),
        }
        # parity of the deterministic rows (and of decryption): GPU public API == the composition on the CPU port
        c0 = next(iter(capis.values()))
        assert sk.decrypt(ct_dec) == c0.decrypt(*h_dec) == [float(v) for v in x_dec], "reference_bench: decrypt parity"
        for name, got in (("BM_Add_CTCT", ct_x + ct_y), ("BM_Add_CTPT", ct_xx + y), ("BM_Mul_CTPT", ct_x * y)):
            want_w, want_e = cpu_f[name](c0)
            got_w, got_e = host(got)
            if not (np.array_equal(got_w, want_w) and list(got_e) == list(want_e)):
                raise SystemExit(f"bench.py: reference_bench parity failed for {name}/{nb}")
        if not np.array_equal(host(ct_x * x)[0], h_xx[0]):
            raise SystemExit("bench.py: reference_bench ct * x is not deterministic")
        for name, (gpu_call, _) in cases.items():
            row = {"gpu_api_us": us(gpu_call, sync=True)}
            if name.startswith("BM_Add"):
                # an addition returns a lazily tagged ciphertext (one Montgomery product); the reference and the CPU leg return
                # the fully reduced wire form, so the row compared with them pays the retag product inside the timed call
                row["gpu_api_lazy_us"] = row["gpu_api_us"]
                row["gpu_api_us"] = us(lambda: gpu_call().words, sync=True)
            for label, c in capis.items():
                row[label] = us(lambda: cpu_f[name](c), sync=False)
            best_cpu = min(v for k_, v in row.items() if k_.startswith("cpu_"))
            row["gpu_over_best_cpu"] = row["gpu_api_us"] / best_cpu
            out["rows"][f"{name}/{nb}"] = row
    out["parity_checked"] = "decrypt values; Add_CTCT / Add_CTPT / Mul_CTPT ciphertext bits and exponents (GPU API vs CPU port), 16 and 64"
    out["note"] = ("gpu_api_us: PaillierPublicKey / PaillierPrivateKey / PaillierEncryptedNumber calls, host ndarray in, each followed "
                   "by torch.cuda.synchronize(); BM_Add_* rows time the addition INCLUDING its export to the wire form (.words), the "
                   "work the reference and the CPU leg do; gpu_api_lazy_us is the addition alone (lazily tagged result).  cpu_*: c_oracle.CApi, Python glue of the reference's shape included")
    return out


def _self_launch(args) -> None:
    """`python bench.py --gpus N` outside a torchrun environment: start the N ranks (one process per GPU) under
    torch.distributed.run on 127.0.0.1 and hand its exit code back.  Fails loudly when the box has fewer GPUs."""
    import socket
    import subprocess

    import torch

    backend = os.environ.get("PAI_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if ndev < args.gpus and backend == "nccl":
        raise SystemExit(f"bench.py: --gpus {args.gpus} but only {ndev} GPU(s) are visible (RCCL needs one device per rank; "
                         f"PAI_BENCH_BACKEND=gloo shares devices for plumbing tests only)")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), PAI_BENCH_SELF_LAUNCHED="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr",
           "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", choices=tuple(CONFIGS), default="headline",
                    help="BASELINE.json configuration: headline (2048-bit, 2^20), cfg2 (2048-bit, 65 536), cfg4 (3072-bit, 2^20), "
                         "cfg5 (4096-bit, 2^18)")
    ap.add_argument("--key-bits", type=int, default=None, choices=(1024, 2048, 3072, 4096), help="override the configuration's key size")
    ap.add_argument("--batch", type=int, default=None, help="elements in total (strong) or per GPU (weak) per step "
                                                            "(default: the configuration's batch)")
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong")
    ap.add_argument("--no-other-scaling", action="store_true", help="N > 1: skip the second (weak resp. strong) arrangement")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip api_level / other_ops / small_batch (profiling runs)")
    ap.add_argument("--no-reference-bench", action="store_true", help="skip the reproduction of bench/bench_ipcl_python.py")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU-baseline sample duration")
    ap.add_argument("--no-configs", action="store_true", help="skip the configs block (BASELINE configs[1], [3], [4] after the headline leg)")
    ap.add_argument("--config-steps", type=int, default=2, help="timed steps of each configuration in the configs block")
    ap.add_argument("--config-cpu-seconds", type=float, default=3.0, help="CPU sample duration of each configuration in the configs block")
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    KEY_BITS = args.key_bits or cfg["key_bits"]
    if args.batch is None:
        args.batch = cfg["batch"]
    CANON_MAC_ENC, CANON_MAC_DEC, CANON_MAC_ADD, CANON_MAC_MUL53 = CANON[KEY_BITS]
    BYTES_ENC, BYTES_DEC, BYTES_ADD = alg_bytes(KEY_BITS)
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        _self_launch(args)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible and there is no CPU fallback")
    # PAI_BENCH_BACKEND=gloo (plumbing test on a box with fewer GPUs than ranks): ranks share the visible devices and the
    # collectives go through the host; the product path is the same
    backend = os.environ.get("PAI_BENCH_BACKEND", "nccl")
    if backend == "nccl" and torch.cuda.device_count() < world:
        raise SystemExit(f"bench.py: {world} ranks but only {torch.cuda.device_count()} GPU(s) visible")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    ranks_seen = [{"rank": 0, "device": dev_index, "name": torch.cuda.get_device_name(dev_index)}]
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
        assert dist.get_world_size() == args.gpus
        # every rank reports through the collective backend itself (RCCL on GPUs): N distinct ranks must answer
        ids = [torch.zeros(2, dtype=torch.int64, device=device) for _ in range(world)]
        dist.all_gather(ids, torch.tensor([rank, dev_index], dtype=torch.int64, device=device))
        ranks_seen = [{"rank": int(t[0]), "device": int(t[1])} for t in ids]
        if sorted(r_["rank"] for r_ in ranks_seen) != list(range(world)):
            raise SystemExit(f"bench.py: the {backend} communicator reports ranks {ranks_seen}, expected {world} distinct ranks")
        if backend == "nccl" and len({r_["device"] for r_ in ranks_seen}) != world:
            raise SystemExit(f"bench.py: ranks share devices under RCCL: {ranks_seen}")

    from pailliercryptolib_python_amd import engine, fixedpoint, sharding

    # the oracle enters only as the checker of what was timed and as the CPU baseline
    from oracle import paillier_oracle as orc

    def make_ctx(bits: int) -> SimpleNamespace:
        """Key material, device handles and the checker's key for one key size."""
        key_ = synthetic_key(bits, DJN_X)
        pub_ = engine.PublicKeyHandle(key_.n, bits, key_.hs, key_.randbits, device=device)
        priv_ = engine.PrivateKeyHandle(pub_, key_.p, key_.q)
        okey_ = orc.make_key(key_.p, key_.q, djn_x=DJN_X, bits=bits)
        assert okey_.n == key_.n and okey_.hs == key_.hs and okey_.randbits == key_.randbits
        return SimpleNamespace(bits=bits, key=key_, pub=pub_, priv=priv_, okey=okey_)

    ctx0 = make_ctx(KEY_BITS)
    key, pub, priv, okey = ctx0.key, ctx0.pub, ctx0.priv, ctx0.okey

    def barrier():
        if world > 1:
            dist.barrier()

    def run_leg(scaling: str, ctx: Optional[SimpleNamespace] = None, batch: Optional[int] = None, steps: Optional[int] = None,
                warmup: Optional[int] = None):
        """Inputs of this rank for one arrangement, W warm-up steps, K timed steps (barrier + synchronize on both sides,
        maximum over ranks), then the parity check of what was timed."""
        ctx = ctx or ctx0
        key, pub, priv, okey = ctx.key, ctx.pub, ctx.priv, ctx.okey
        batch = args.batch if batch is None else batch
        steps = args.steps if steps is None else steps
        warmup = args.warmup if warmup is None else warmup
        if scaling == "strong":
            begin, B = engine.shard_plan(batch, world)[rank]             # this rank's contiguous block of the global batch
            x = np.random.default_rng(1002).uniform(-1000.0, 1000.0, batch)[begin:begin + B]
            total_per_step = float(batch)
        else:
            begin, B = 0, batch
            x = np.random.default_rng(1002 + rank).uniform(-1000.0, 1000.0, B)
            total_per_step = float(B) * world
        res, expo = fixedpoint.encode_float64_array(x, key.n, pub.n_words)
        m = engine.to_device_words(res, device)
        gen = torch.Generator(device=device)
        gen.manual_seed(4002 + rank)
        r = pub.random_r(max(B, 1), generator=gen)[:B].contiguous()
        ct = pub.empty_ct(B)
        out = pub.empty_pt(B)
        torch.cuda.synchronize()

        def step():
            if B:
                pub.encrypt(m, r, out=ct)
                priv.decrypt(ct, out=out)

        for _ in range(warmup):
            step()
        torch.cuda.synchronize()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        barrier()
        elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())

        # ---- correctness of what was just timed (outside the timed region) ---------------------------
        ok = bool(torch.equal(out, m))
        nchk = min(B, 4096)
        got_x = fixedpoint.decode_float64_array(engine.to_host_words(out[:nchk]), expo[:nchk], key.n, key.max_int)
        ok = ok and bool(np.array_equal(got_x, x[:nchk]))
        if B:
            idx = sorted({0, min(1, B - 1), B // 2, B - 1})
            ct_h = engine.to_host_words(ct[idx])
            r_h = engine.words_to_ints(engine.to_host_words(r[idx]))
            m_h = engine.words_to_ints(res[idx])
            ok = ok and engine.words_to_ints(ct_h) == [orc.encrypt(okey, mm, rr) for mm, rr in zip(m_h, r_h)]
        if world > 1:
            flag = torch.tensor([1 if ok else 0], device=device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = bool(flag.item())
        if not ok:
            raise SystemExit("bench.py: parity check failed (decrypt(encrypt(m)) != m or ciphertext bits differ from the oracle)")
        return SimpleNamespace(scaling=scaling, begin=begin, B=B, x=x, res=res, expo=expo, m=m, r=r, ct=ct, out=out,
                               elapsed=elapsed, total_per_step=total_per_step, steps=steps, warmup=warmup)

    def kernel_times(ctx: SimpleNamespace, leg_: SimpleNamespace, reps: int) -> dict:
        """Per-kernel durations (ms) of one step, HIP events on the launch stream, mean of `reps` passes."""
        engine.profile_enable(True)
        acc_ = {}
        for _ in range(reps):
            if not leg_.B:
                break
            ctx.pub.encrypt(leg_.m, leg_.r, out=leg_.ct)
            for k_, v in engine.profile_last().items():
                acc_[k_] = acc_.get(k_, 0.0) + v
            ctx.priv.decrypt(leg_.ct, out=leg_.out)
            for k_, v in engine.profile_last().items():
                acc_[k_] = acc_.get(k_, 0.0) + v
        engine.profile_enable(False)
        return {k_: v / reps for k_, v in acc_.items()}

    leg = run_leg(args.scaling)
    begin, B, x, res, expo, m, r, ct, out = leg.begin, leg.B, leg.x, leg.res, leg.expo, leg.m, leg.r, leg.ct, leg.out
    elapsed, total_per_step = leg.elapsed, leg.total_per_step

    # ---- the final gather of the sharded result (strong scaling; RCCL all-gather over xGMI) ------
    gather_ms = None
    if args.scaling == "strong" and world > 1:
        full = sharding.gather_rows(ct, args.batch)              # warm-up (communicator set-up)
        torch.cuda.synchronize()
        barrier()
        t1 = time.perf_counter()
        full = sharding.gather_rows(ct, args.batch)
        torch.cuda.synchronize()
        gather_ms = 1e3 * (time.perf_counter() - t1)
        if not torch.equal(full[begin:begin + B], ct):
            raise SystemExit("bench.py: gathered ciphertexts differ from the local shard")
        del full

    # ---- per-kernel durations with HIP events on the launch stream (every rank; rank 0 reports) ----
    kern = kernel_times(ctx0, leg, max(1, min(args.steps, 3)))
    per_rank_kern = [kern]
    if world > 1:
        per_rank_kern = [None] * world
        dist.all_gather_object(per_rank_kern, kern)

    # ---- N > 1: the other arrangement, timed the same way (same K, same barriers, same parity check) ----
    other_leg = None
    if world > 1 and not args.no_other_scaling:
        o = run_leg("weak" if args.scaling == "strong" else "strong")
        other_leg = {"scaling": o.scaling, "value": o.total_per_step * args.steps / o.elapsed, "unit": "ops/s",
                     "ms_per_step": 1e3 * o.elapsed / args.steps, "batch_total": int(o.total_per_step),
                     "batch_this_rank": o.B, "steps": args.steps, "parity_checked": True}
        del o

    single = rank == 0 and world == 1
    extras = single and not args.no_extras

    # ---- CPU baseline on the host cores (rank 0, N = 1 only) -------------------------------------
    def cpu_baseline_for(ctx: SimpleNamespace, leg_: SimpleNamespace, seconds: float, min_sample: int, with_small: bool):
        """The same two operations on the host cores, on a bounded sample of the batch that was just timed (same key, same
        randomness; ciphertext bits compared with the GPU's)."""
        import ctypes.util

        from oracle import c_oracle as co

        ck = co.COracleKey(ctx.okey)
        # the reference's own kernel library, if the box happens to have it
        mb_lib = ctypes.util.find_library("crypto_mb") or ctypes.util.find_library("ippcp")
        if co.ifma_available():
            enc, dec, how, kind = ck.ifma_encrypt_djn, ck.ifma_decrypt_crt, "AVX512-IFMA mb8 port oracle/paillier_ifma.c (8 lanes x 52-bit limbs, 5-bit windows)", "port (IFMA mb8)"
        elif co.gmp_available():
            enc, dec, how, kind = ck.gmp_encrypt_djn, ck.gmp_decrypt_crt, "libgmp mpz_powm via oracle/paillier_ref.c", "port (GMP)"
        else:
            enc, dec, how, kind = ck.encrypt_djn, ck.decrypt_crt, "plain-C CIOS port oracle/paillier_ref.c", "port"
        threads = co.max_threads()
        res_, r_, ct_ = leg_.res, leg_.r, leg_.ct
        cap = min(leg_.B, 8192 * max(1, threads // 2))
        r_host = engine.to_host_words(r_[:cap])
        probe = min(cap, 8 * threads)
        dec(enc(res_[:probe], r_host[:probe]))                       # cold call: thread pool, page faults
        t1 = time.perf_counter()
        dec(enc(res_[:probe], r_host[:probe]))                       # warm probe sizes the sample
        t_probe = time.perf_counter() - t1
        sample = int(min(cap, max(min_sample, probe * seconds / max(t_probe, 1e-4))))
        sample -= sample % 8
        t1 = time.perf_counter()
        c_ct = enc(res_[:sample], r_host[:sample])
        t_enc_cpu = time.perf_counter() - t1
        c_m = dec(c_ct)
        t_cpu = time.perf_counter() - t1
        assert np.array_equal(c_m, res_[:sample]), "CPU baseline failed its own round trip"
        nchk_ = min(256, sample)
        assert np.array_equal(c_ct[:nchk_], engine.to_host_words(ct_[:nchk_])), "CPU baseline and GPU ciphertexts differ"
        cpu_ = {
            "value": sample / t_cpu, "unit": "encrypt+decrypt ops/s", "cores": threads, "kind": kind,
            "sample": f"{sample} elements of the same batch, same key and randomness; {how}, OpenMP over {threads} host "
                      f"threads, {t_cpu:.1f} s (encrypt {t_enc_cpu:.1f} s)",
            "cores_detail": {"used": threads, "host_total": os.cpu_count(),
                             "note": "used = min(OpenMP default, CPU affinity, cgroup cpu.max quota) of this process"},
            "reference_kernel_library": mb_lib or "libcrypto_mb / libippcp not present on this box (IPP-Crypto is un-vendored upstream)",
        }
        # the reference's own benchmark sizes (bench/bench_ipcl_python.py:24-25,34-35,45-46,56-57,67-68: 16 and 64 elements),
        # same port, one thread and all threads, beside the GPU latencies of small_batch
        if with_small and co.ifma_available():
            def cpu_wall(f, reps=5):
                f()
                ts_ = []
                for _ in range(reps):
                    t1_ = time.perf_counter()
                    f()
                    ts_.append(time.perf_counter() - t1_)
                return 1e3 * sorted(ts_)[len(ts_) // 2]            # median of 5 (one descheduled run does not move it)

            e53_h = [int(v) | 1 << 52 for v in np.random.default_rng(5).integers(0, 1 << 52, 64)]
            cpu_small = {"kind": kind, "unit": "ms per batch", "threads_all": threads}
            for nb in (16, 64):
                ct_s = c_ct[:nb]
                cpu_small[str(nb)] = {}
                for label, th in (("1_thread", 1), ("all_threads", threads)):
                    cpu_small[str(nb)][label] = {
                        "encrypt_ms": cpu_wall(lambda: enc(res_[:nb], r_host[:nb], threads=th)),
                        "decrypt_ms": cpu_wall(lambda: dec(ct_s, threads=th)),
                        "ct_add_ms": cpu_wall(lambda: co.modmul(ctx.key.nsq, ct_s, ct_s, threads=th)),
                        "ct_mul_53bit_ms": cpu_wall(lambda: co.ifma_modexp(ctx.key.nsq, ct_s, e53_h[:nb], threads=th)),
                    }
            cpu_["small_batch"] = cpu_small
        return cpu_

    cpu = None
    if single and not args.no_cpu_baseline:
        cpu = cpu_baseline_for(ctx0, leg, args.cpu_seconds, 8192, True)

    # ---- API level: host float64 -> encrypt -> decrypt -> host float64 (rank 0, N = 1) ----------
    api = None
    if extras:
        from pailliercryptolib_python_amd import PaillierPrivateKey, PaillierPublicKey
        from pailliercryptolib_python_amd.bindings import ipclPublicKey

        apk = PaillierPublicKey(ipclPublicKey(key.n, KEY_BITS, True, hs=key.hs, randbits=key.randbits, device=device))
        ask = PaillierPrivateKey(apk, key.p, key.q)
        def api_pass():
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            en_ = apk.encrypt(x)
            torch.cuda.synchronize()
            te = time.perf_counter() - t1
            t1 = time.perf_counter()
            back_ = ask.decrypt_to_numpy(en_)
            td = time.perf_counter() - t1
            if not np.array_equal(back_, x):
                raise SystemExit("bench.py: API-level round trip failed")
            return te, td

        first = api_pass()                   # first call of this size: fixed-base table build, scratch growth
        t_api_enc, t_api_dec = api_pass()    # steady state
        api = {"encrypt_s": t_api_enc, "decrypt_s": t_api_dec, "ops_per_s": B / (t_api_enc + t_api_dec), "batch": B,
               "first_call": {"encrypt_s": first[0], "decrypt_s": first[1]},
               "note": "PaillierPublicKey.encrypt(float64 ndarray) + PaillierPrivateKey.decrypt_to_numpy, steady state (second "
                       "full-size call; first_call includes the one-off fixed-base table build and scratch allocation): "
                       "device codec, ChaCha20 randomness under an OS-CSPRNG key, H2D/D2H over PCIe included"}

    # ---- the other operations of BASELINE configs[2], kernel-resident, each checked against the oracle ----
    other = None
    small = None
    if extras:
        def wall(f, reps=3):
            f()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(reps):
                f()
            torch.cuda.synchronize()
            return (time.perf_counter() - t1) / reps

        chk = sorted({int(v) for v in np.linspace(0, B - 1, 64)})      # 64 elements spread over the batch

        def rows(t_):
            return engine.words_to_ints(engine.to_host_words(t_[chk]))

        NLSQ = modmul_limbs(2 * KEY_BITS)                 # limbs of the lane-group geometry serving n^2
        ct_b = torch.roll(ct, 1, dims=0).contiguous()
        ct2 = pub.empty_ct(B)
        # dense 53-bit multipliers (float mantissas): all 32 bits of the low word and the 20 below the top bit are random
        e53 = torch.randint(-(1 << 31), 1 << 31, (B, 2), dtype=torch.int64, device=device).to(torch.int32)
        e53[:, 1] &= (1 << 21) - 1
        e53[:, 1] |= 1 << 20
        ca, cb = rows(ct), rows(ct_b)
        t_add = wall(lambda: pub.ct_add(ct, ct_b, out=ct2))
        if rows(ct2) != [orc.ct_add(a, b, key.nsq) for a, b in zip(ca, cb)]:
            raise SystemExit("bench.py: ct_add parity check failed")
        t_add_b = wall(lambda: pub.ct_add(ct, ct_b[:1], out=ct2))
        if rows(ct2) != [orc.ct_add(a, engine.words_to_ints(engine.to_host_words(ct_b[:1]))[0], key.nsq) for a in ca]:
            raise SystemExit("bench.py: broadcast ct_add parity check failed")
        # the same addition inside a chain (lazy Montgomery domain, include/paillier_hip.h pai_ct_mont_mul): ONE product, the
        # stray R^-1 kept as the buffer's tag; checked as a b R^-1 and, after the retag product, as a b (the wire form)
        t_add_1 = wall(lambda: pub.ct_mont_mul(ct, ct_b, out=ct2))
        r_inv = pow(pow(2, pub.mont_bits, key.nsq), -1, key.nsq)
        if rows(ct2) != [a * b * r_inv % key.nsq for a, b in zip(ca, cb)]:
            raise SystemExit("bench.py: ct_mont_mul parity check failed")
        t_retag = wall(lambda: pub.ct_retag(ct2, -1, 0, out=ct2), reps=1)
        pub.ct_mont_mul(ct, ct_b, out=ct2)
        pub.ct_retag(ct2, -1, 0, out=ct2)
        if rows(ct2) != [orc.ct_add(a, b, key.nsq) for a, b in zip(ca, cb)]:
            raise SystemExit("bench.py: ct_mont_mul + retag parity check failed")
        t_mul = wall(lambda: pub.ct_mul(ct, e53, 53, out=ct2), reps=2)
        e_h = [int(v[0]) & 0xFFFFFFFF | (int(v[1]) & 0xFFFFFFFF) << 32 for v in e53[chk].cpu().numpy()]
        if rows(ct2) != [orc.ct_mul(a, e, key.nsq) for a, e in zip(ca, e_h)]:
            raise SystemExit("bench.py: ct_mul parity check failed")
        t_inv = wall(lambda: pub.ct_invert(ct, out=ct2), reps=2)
        if rows(ct2) != [orc.ct_inv(a, key.nsq) for a in ca]:
            raise SystemExit("bench.py: ct_invert parity check failed")
        nsum = min(B, 1 << 16)
        t_sum = wall(lambda: pub.ct_prod(ct[:nsum], 1), reps=2)
        prod_h = engine.words_to_ints(engine.to_host_words(pub.ct_prod(ct[:4096].contiguous(), 1)))[0]
        want = 1
        for v in engine.words_to_ints(engine.to_host_words(ct[:4096])):
            want = want * v % key.nsq
        if prod_h != want:
            raise SystemExit("bench.py: ct_prod parity check failed")
        engine.profile_enable(True)
        pub.ct_add(ct, ct_b, out=ct2)
        k_add_all = dict(engine.profile_last())
        k_add_name = next((k for k in ("k_modmul_msb", "k_modmul") if k in k_add_all), None)
        k_add = k_add_all.get(k_add_name)
        engine.profile_enable(False)
        # wire-form additions of large batches: ONE most-significant-limb-first pass of 2 NL^2 limb products + 2 per row for the
        # quotient digit (csrc/mont_msb.hpp) where the key has the context, else two Montgomery products
        macs_add_wire = (2 * NLSQ * NLSQ + 2 * NLSQ) if k_add_name == "k_modmul_msb" else 2 * 2 * NLSQ * NLSQ
        other = {
            "ct_add_ops_per_s": B / t_add, "ct_add_bcast_ops_per_s": B / t_add_b, "ct_mul_53bit_ops_per_s": B / t_mul,
            "ct_add_in_chain_ops_per_s": B / t_add_1, "ct_retag_ops_per_s": B / t_retag,
            "ct_add_in_chain_roofline": {"bound": "valu_int", "canonical_frac": CANON_MAC_ADD * B / t_add_1 / PEAK_MAC32_PER_S,
                                         "executed_frac": 2 * NLSQ * NLSQ * B / t_add_1 / PEAK_MAC32_PER_S,
                                         "hbm_GBs": BYTES_ADD * B / t_add_1 / 1e9,
                                         "note": "one Montgomery product per addition; the wire form costs one more product "
                                                 "(ct_retag) once per chain, at the boundary"},
            "ct_invert_ops_per_s": B / t_inv, "ct_sum_elements_per_s": nsum / t_sum, "batch": B,
            "ct_add_kernel_ms": k_add, "ct_add_kernel": k_add_name,
            "ct_add_roofline": {"bound": "valu_int", "canonical_frac": CANON_MAC_ADD * B / t_add / PEAK_MAC32_PER_S,
                                "executed_macs_per_element": macs_add_wire,
                                "executed_frac": macs_add_wire * B / t_add / PEAK_MAC32_PER_S,
                                "hbm_GBs": BYTES_ADD * B / t_add / 1e9, "hbm_frac": BYTES_ADD * B / t_add / 1e9 / HBM_PEAK_GBS},
            # k_ctmul_padic on 72-limb base-n digit pairs, 53-bit exponents, 3-bit windows: 52 squarings of 4 NL^2, ~18 window
            # products + 6 table products + 4 conversion products of 5 NL^2 (canonical: SURVEY 8d's 3.16 M MAC32)
            "ct_mul_roofline": {"bound": "valu_int", "canonical_frac": 3.16e6 * B / t_mul / PEAK_MAC32_PER_S,
                                "executed_frac": ((52 * 4 + 28 * 5) * 72 * 72 * B / t_mul / PEAK_MAC32_PER_S) if KEY_BITS == 2048 else None},
            "note": "BASELINE configs[2] operations on the same resident batch (wall clock around the C-ABI call); results "
                    "checked against the oracle on 64 elements each",
        }
        del ct_b, ct2, e53
        # latency at the reference's own batch sizes (bench/bench_ipcl_python.py:24-25,34-35,45-46)
        small = {}
        for nb in (16, 64) + ((8192,) if B >= 8192 else ()):         # 8 192: a mid-size batch (lane-group digit pairs, 4 lanes per chain)
            ms_, rs_ = m[:nb].contiguous(), r[:nb].contiguous()
            cts_ = pub.encrypt(ms_, rs_)
            if nb > 64 and not (torch.equal(cts_, ct[:nb]) and torch.equal(priv.decrypt(cts_), ms_)):
                raise SystemExit("bench.py: the mid-size batch differs from the full batch's ciphertexts / plaintexts")
            # dense random 53-bit exponents: the SAME values the CPU leg of cpu_baseline.small_batch uses (default_rng(5))
            e53_s = [int(v) | 1 << 52 for v in np.random.default_rng(5).integers(0, 1 << 52, max(64, nb))][:nb]
            es_ = torch.from_numpy(np.array([[e & 0xFFFFFFFF, e >> 32] for e in e53_s], dtype=np.int64).astype(np.uint32)
                                   .view(np.int32)).to(device)
            small[str(nb)] = {
                "encrypt_ms": 1e3 * wall(lambda: pub.encrypt(ms_, rs_), reps=5),
                "decrypt_ms": 1e3 * wall(lambda: priv.decrypt(cts_), reps=5),
                "ct_add_ms": 1e3 * wall(lambda: pub.ct_add(cts_, cts_), reps=5),
                "ct_mul_53bit_ms": 1e3 * wall(lambda: pub.ct_mul(cts_, es_, 53), reps=5),
            }

    # ---- DJN encryption at the two table operating points: the big table a device's first keys
    # get, and the small one a handle takes when many keys are resident — a second handle of the same key, forced small
    table_points = None
    if extras:
        def enc_ms(h_):
            h_.encrypt(m, r, out=ct)
            engine.profile_enable(True)
            h_.encrypt(m, r, out=ct)
            ms_ = engine.profile_last().get("k_encrypt(djn)")
            engine.profile_enable(False)
            return ms_

        big = dict(pub.table_info(), k_encrypt_ms=enc_ms(pub), batch=B)
        os.environ["PAI_FB_BIG_KEYS"] = "0"                      # every further key of this process: the small operating point
        try:
            pub_s = engine.PublicKeyHandle(key.n, KEY_BITS, key.hs, key.randbits, device=device)
            ct_small = pub.empty_ct(B)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            pub_s.encrypt(m, r, out=ct_small)
            torch.cuda.synchronize()
            first_s = time.perf_counter() - t1
            if not torch.equal(ct_small, ct):
                raise SystemExit("bench.py: the small-table encryption differs from the big-table one")
            small_pt = dict(pub_s.table_info(), k_encrypt_ms=enc_ms(pub_s), first_call_s=first_s, batch=B)
            del pub_s, ct_small
        finally:
            os.environ.pop("PAI_FB_BIG_KEYS", None)
        table_points = {"big": big, "small": small_pt,
                        "note": "same key, same ciphertext bits; big = the table of a device's first PAI_FB_BIG_KEYS (8) keys, small = "
                                "what a handle builds when that many tables are resident or the cache budget is short "
                                "(PAI_FB_SMALL_TABLE_MB, default 256); first_call_s includes the table build"}

    # ---- the standard scheme (enable_DJN=False, ipcl_python.py:20-40; classes.cpp:24-27): ct = (1 + m n) r^n mod n^2 with a
    # full-size r — r^n on base-n digit pairs (k_pow_padic) and one fused product; same batch, oracle bits on 64 samples
    std_scheme = None
    if extras and padic_nl(KEY_BITS):
        pub_std = engine.PublicKeyHandle(key.n, KEY_BITS, None, 0, device=device)
        gen_s = torch.Generator(device=device)
        gen_s.manual_seed(977)
        r_std = torch.randint(-(1 << 31), 1 << 31, (B, pub_std.r_words), dtype=torch.int64, device=device, generator=gen_s).to(torch.int32)
        r_std[:, -1] &= (1 << ((key.n.bit_length() - 1) % 32)) - 1 if (key.n.bit_length() - 1) % 32 else 0    # r < 2^(bits-1) < n
        ct_std = pub.empty_ct(B)
        pub_std.encrypt(m, r_std, out=ct_std)                     # first call: schedule, scratch
        torch.cuda.synchronize()
        engine.profile_enable(True)
        t1 = time.perf_counter()
        pub_std.encrypt(m, r_std, out=ct_std)
        torch.cuda.synchronize()
        t_std = time.perf_counter() - t1
        k_std = dict(engine.profile_last())
        engine.profile_enable(False)
        okey_std = orc.make_key(key.p, key.q, djn_x=None, bits=KEY_BITS)
        chk_s = sorted({int(v) for v in np.linspace(0, B - 1, 64)})
        want_s = [orc.encrypt(okey_std, mm, rr) for mm, rr in zip(engine.words_to_ints(res[chk_s]),
                                                                  engine.words_to_ints(engine.to_host_words(r_std[chk_s])))]
        if engine.words_to_ints(engine.to_host_words(ct_std[chk_s])) != want_s:
            raise SystemExit("bench.py: standard-scheme ciphertext bits differ from the oracle")
        if not torch.equal(priv.decrypt(ct_std), m):
            raise SystemExit("bench.py: standard-scheme round trip failed")
        macs_std = executed_macs_std_obfuscator(key.n)
        std_scheme = {"encrypt_ops_per_s": B / t_std, "encrypt_ms": 1e3 * t_std, "batch": B, "key_bits": KEY_BITS, "kernel_ms": k_std,
                      "executed_macs_per_element": macs_std, "executed_frac": macs_std * B / t_std / PEAK_MAC32_PER_S,
                      "oracle_samples": len(chk_s), "round_trip_checked": True,
                      "note": "enable_DJN=False: (1 + m n) r^n mod n^2, r of the key's size; r^n = sliding-window power on base-n "
                              "digit pairs (squarings 4 NL^2, products 5 NL^2, NL = 72) + one fused product; bits against the "
                              "oracle, decrypt(ct) == m on the whole batch"}
        del pub_std, r_std, ct_std

    ref_bench = None
    if extras and KEY_BITS == 2048 and not args.no_reference_bench:
        ref_bench = reference_bench(key, okey, device)

    def roofline_for(ctx: SimpleNamespace, B_: int, kern_: dict) -> dict:
        """The dominant kernel (k_dec_a_padic: both CRT half-size exponentiations) against the integer-VALU roof."""
        k_ = ctx.key
        canon_enc, canon_dec = CANON[ctx.bits][0], CANON[ctx.bits][1]
        bytes_enc, bytes_dec, _ = alg_bytes(ctx.bits)
        t_deca = kern_.get("k_dec_a", 0.0) * 1e-3
        t_enc = kern_.get("k_encrypt(djn)", 0.0) * 1e-3
        macs_exec = executed_macs_decrypt(k_.p, k_.q)
        canonical = (canon_dec * B_ / t_deca) if t_deca > 0 else None
        executed = (macs_exec * B_ / t_deca) if t_deca > 0 else None
        traffic = pmc_traffic("k_dec_a_padic", B_, ctx.bits)
        return {
            "bound": "valu_int",
            "kernel": f"k_dec_a_padic<{padic_nl(max(k_.p.bit_length(), k_.q.bit_length()))}> (CRT-decrypt stage A: (ct mod s^2)^(s-1) for both primes)",
            "achieved": (executed / 1e12) if executed else None,
            "peak": PEAK_MAC32_PER_S / 1e12,
            "peak_sustained": PEAK_SUSTAINED_MAC32_PER_S / 1e12,
            "frac_of_sustained": (executed / PEAK_SUSTAINED_MAC32_PER_S) if executed else None,
            "peak_note": "peak = 18 ms burst, constant operands (profiles/r01/ubench_valu_mi355x.jsonl); peak_sustained = 3 s of "
                         "back-to-back launches on data-dependent operands at the clock power management settles at (2.29 GHz, "
                         "1290 W of 1400 W): profiles/r04/ubench_valu_sustained.jsonl + _power.txt.  One wave per SIMD (this "
                         "kernel's occupancy: LDS-bound) issues the same instruction stream at 26.7 T MAC/s",
            "unit": "T MAC/s (29x29-bit multiply-accumulates actually executed, v_mad_u64_u32)",
            "frac": (executed / PEAK_MAC32_PER_S) if executed else None,
            "executed_macs_per_element": macs_exec, "canonical_mac32_per_element": canon_dec,
            "canonical_T_MAC32_s": (canonical / 1e12) if canonical else None,
            "canonical_frac": (canonical / PEAK_MAC32_PER_S) if canonical else None,
            "note": "frac = executed MACs / measured v_mad_u64_u32 peak (kernel quality).  canonical_* price the kernel at "
                    "the CANONICAL algorithm's work (SURVEY §8d: CIOS mod s^2, 5-bit windows); the kernel runs a cheaper "
                    "algorithm (arithmetic mod s on base-s digit pairs), so the canonical fraction can exceed 1",
            "kernel_ms": kern_,
            "traffic": traffic["bytes"] if traffic else None,
            "traffic_source": traffic,
            "traffic_unit": "HBM bytes per k_dec_a_padic launch (PMC: 2 x FETCH_SIZE + WRITE_SIZE, scaled from the profile's batch)",
            "hbm": {
                "bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                "achieved_decrypt": (bytes_dec * B_ / t_deca / 1e9) if t_deca > 0 else None,
                "achieved_encrypt": (bytes_enc * B_ / t_enc / 1e9) if t_enc > 0 else None,
            },
        }

    # ---- the other BASELINE.json configurations, driver-timed in the same run (rank 0, N = 1, default invocation) ----
    # configs[1] (2048-bit, 65 536), configs[3] (3072-bit, 2^20), configs[4] (4096-bit, 2^18): the same step, the same
    # barrier / synchronize bracket, the same full-batch parity check (decrypt(encrypt(m)) == m on every element, ciphertext
    # bits against the oracle on samples), per-kernel HIP-event times, the same roofline object and a CPU sample each
    configs_block = None
    if single and args.config == "headline" and args.key_bits is None and args.batch == CONFIGS["headline"]["batch"] \
            and not args.no_configs:
        configs_block = {}
        for name in ("cfg2", "cfg4", "cfg5"):
            c_ = CONFIGS[name]
            ctx_ = ctx0 if c_["key_bits"] == KEY_BITS else make_ctx(c_["key_bits"])
            t_first = time.perf_counter()
            leg_ = run_leg("strong", ctx_, c_["batch"], args.config_steps, 1)
            t_first = time.perf_counter() - t_first
            kern_ = kernel_times(ctx_, leg_, 1)
            blk = {
                "baseline_config_is": c_["baseline"], "key_bits": c_["key_bits"], "batch": c_["batch"],
                "value": leg_.total_per_step * leg_.steps / leg_.elapsed, "unit": "ops/s",
                "ms_per_step": 1e3 * leg_.elapsed / leg_.steps, "steps": leg_.steps, "warmup": leg_.warmup,
                "wall_s_including_key_setup_table_build_and_parity_check": t_first,
                "parity_checked": True, "roofline": roofline_for(ctx_, leg_.B, kern_),
                "cpu_baseline": None if args.no_cpu_baseline else cpu_baseline_for(ctx_, leg_, args.config_cpu_seconds, 512, False),
            }
            configs_block[name] = blk
            del leg_
            if ctx_ is not ctx0:
                ctx_.pub.trim()
                del ctx_
            torch.cuda.empty_cache()

    if rank == 0:
        total_ops = total_per_step * args.steps
        value = total_ops / elapsed
        metric = BASELINE_METRIC if args.config == "headline" and KEY_BITS == 2048 else \
            f"Paillier encrypt+decrypt ops/sec, {KEY_BITS}-bit key, batch={args.batch}; {world} MI355X ({cfg['baseline']})"
        line = {
            "metric": metric,
            "value": value,
            "unit": "ops/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "u32 limbs (radix-2^29 in 32-bit registers, 64-bit accumulators)",
            "data": "synthetic",
            "config": {
                "workload": f"{KEY_BITS}-bit key ({'reference bench P,Q' if KEY_BITS == 2048 else 'seeded fixture primes'}), DJN encrypt + CRT decrypt, "
                            + (f"batch={args.batch} in total, block-sharded over {world} GPU(s)" if args.scaling == "strong"
                               else f"batch={B} per GPU")
                            + ", inputs resident in HBM",
                "baseline_config": args.config, "baseline_config_is": cfg["baseline"],
                "key_bits": KEY_BITS, "batch_per_gpu": B if args.scaling == "weak" else None,
                "batch_total": int(total_per_step), "scheme": "DJN", "parallelism": f"shard{world}",
            },
            "roofline": roofline_for(ctx0, B, kern),
            "per_rank_kernel_ms": per_rank_kern if world > 1 else None,
            "gather_ms": gather_ms,
            "ranks_seen": ranks_seen,
            "collective_backend": (backend if world > 1 else None),
            ("weak_scaling" if args.scaling == "strong" else "strong_scaling"): other_leg,
            "cpu_baseline": cpu,
            "configs": configs_block,
            "api_level": api,
            "other_ops": other,
            "small_batch": small,
            "fixed_base_table_points": table_points,
            "reference_bench": ref_bench,
            "parity_checked": True,
        }
        # ---- the LAST keys of the line (the driver keeps the tail of stdout): every configuration's number, compact ----
        rl = line["roofline"]
        line["roofline"]["note_traffic"] = (
            None if not rl.get("traffic") else
            f"traffic / algorithmic bytes = {rl['traffic'] / (BYTES_DEC * B):.0f}x (per-element window tables in HBM scratch) = "
            f"{rl['traffic'] / max(kern.get('k_dec_a', 0.0) * 1e-3, 1e-9) / 1e9:.0f} GB/s of {HBM_PEAK_GBS:.0f}: not the limiter")
        line["standard_scheme"] = std_scheme
        if ref_bench:
            line["reference_bench_summary_us"] = {
                k_: [round(v_["gpu_api_us"], 1)] + [round(min(x_ for n_, x_ in v_.items() if n_.startswith("cpu_")), 1)
                                                     for _ in (0,) if any(n_.startswith("cpu_") for n_ in v_)]
                for k_, v_ in ref_bench.get("rows", {}).items()}           # [GPU public API, best CPU-port figure]
        line["configs_summary"] = {
            "fields": ["ops_per_s", "ms_per_step", "roofline_frac_executed", "k_encrypt_ms", "k_dec_a_ms", "cpu_ops_per_s"],
            "headline": [round(value), round(1e3 * elapsed / args.steps, 2), rl["frac"] and round(rl["frac"], 4),
                         kern.get("k_encrypt(djn)") and round(kern["k_encrypt(djn)"], 2), kern.get("k_dec_a") and round(kern["k_dec_a"], 2),
                         cpu and round(cpu["value"])],
            **({n_: [round(b_["value"]), round(b_["ms_per_step"], 2), b_["roofline"]["frac"] and round(b_["roofline"]["frac"], 4),
                     b_["roofline"]["kernel_ms"].get("k_encrypt(djn)") and round(b_["roofline"]["kernel_ms"]["k_encrypt(djn)"], 2),
                     b_["roofline"]["kernel_ms"].get("k_dec_a") and round(b_["roofline"]["kernel_ms"]["k_dec_a"], 2),
                     b_["cpu_baseline"] and round(b_["cpu_baseline"]["value"])] for n_, b_ in (configs_block or {}).items()}),
            "standard_scheme_2048": std_scheme and [round(std_scheme["encrypt_ops_per_s"]), round(std_scheme["encrypt_ms"], 1),
                                                    round(std_scheme["executed_frac"], 4)],
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
