"""Dev probe: wire-form ct + ct by ONE most-significant-limb-first product (csrc/mont_msb.hpp, k_modmul_msb) — every element against
CPython at four key sizes (reduced, extreme and unreduced operands), then the timing beside the two Montgomery products and the lazy single
product; `sweep`: small batches on the latency route against the throughput route.   python tools/ctadd_msb_check.py [log2 batch = 20]"""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import torch
from bench import synthetic_key
from pailliercryptolib_python_amd import engine

dev = torch.device('cuda', 0)


def to_ints(t):
    a = t.cpu().numpy().view(np.uint32)
    return [int.from_bytes(r.tobytes(), 'little') for r in a]


def from_ints(vals, words):
    buf = b''.join(int(v).to_bytes(4 * words, 'little') for v in vals)
    return torch.from_numpy(np.frombuffer(buf, dtype=np.uint32).reshape(len(vals), words).view(np.int32).copy()).to(dev)


def check(bits, n_elems, rng):
    key = synthetic_key(bits, 0x1234567 + bits)
    pub = engine.PublicKeyHandle(key.n, bits, key.hs, key.randbits, device=dev)
    M = key.n * key.n
    W = pub.ct_words
    full = (1 << (32 * W)) - 1
    av = [M - 1, 1, 0, M - 1, 1, M // 2, full, full, M - 1, (1 << (M.bit_length() - 1)) - 1]
    bv = [M - 1, 1, 5, 1, M - 1, M - 2, full, M - 1, full, (1 << (M.bit_length() - 1)) - 1]
    while len(av) < n_elems:
        r = len(av) % 4
        if r == 3:      # unreduced words
            av.append(int(rng.integers(0, 2**63)) * full // 2**63); bv.append(int.from_bytes(rng.bytes(4 * W), 'little'))
        else:
            av.append(int.from_bytes(rng.bytes(4 * W), 'little') % M); bv.append(int.from_bytes(rng.bytes(4 * W), 'little') % M)
    a, b = from_ints(av, W), from_ints(bv, W)
    out = pub.empty_ct(n_elems)
    os.environ["PAI_LAT_ADD_MAX"] = "0"                     # lane groups at this batch size too
    engine.profile_enable(True)
    pub.ct_add(a, b, out=out)
    kern = engine.profile_last()
    engine.profile_enable(False)
    os.environ.pop("PAI_LAT_ADD_MAX")
    got = to_ints(out)
    bad = [i for i in range(n_elems) if got[i] != av[i] * bv[i] % M]
    return {"key_bits": bits, "elements": n_elems, "kernel": list(kern), "mismatches": len(bad), "first_bad": bad[:5]}


def timing(bits, B):
    key = synthetic_key(bits, 0x1234567)
    pub = engine.PublicKeyHandle(key.n, bits, key.hs, key.randbits, device=dev)
    g = torch.Generator(device=dev); g.manual_seed(1)
    a = torch.randint(-(2**31), 2**31, (B, pub.ct_words), dtype=torch.int64, device=dev, generator=g).to(torch.int32)
    b = torch.randint(-(2**31), 2**31, (B, pub.ct_words), dtype=torch.int64, device=dev, generator=g).to(torch.int32)
    a[:, -1] &= 0x00FFFFFF; b[:, -1] &= 0x00FFFFFF
    out = pub.empty_ct(B); ref = pub.empty_ct(B)

    def tm(f, reps=5):
        f(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps): f()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3
    row = {"key_bits": bits, "batch": B}
    row["msb_ms"] = round(tm(lambda: pub.ct_add(a, b, out=out)), 3)
    engine.profile_enable(True); pub.ct_add(a, b, out=out); row["msb_kernel"] = engine.profile_last(); engine.profile_enable(False)
    os.environ["PAI_DISABLE"] = "add_msb"
    row["montgomery_x2_ms"] = round(tm(lambda: pub.ct_add(a, b, out=ref)), 3)
    row["lazy_single_product_ms"] = round(tm(lambda: pub.ct_mont_mul(a, b, out=ref)), 3)
    pub.ct_add(a, b, out=ref)
    os.environ.pop("PAI_DISABLE")
    row["same_bits"] = bool(torch.equal(out, ref))
    row["msb_M_per_s"] = round(B / row["msb_ms"] / 1e3, 1)
    row["montgomery_M_per_s"] = round(B / row["montgomery_x2_ms"] / 1e3, 1)
    return row


if __name__ == "__main__":
    lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    rng = np.random.default_rng(7)
    if len(sys.argv) > 2 and sys.argv[2] == "quick":            # the 2048-bit timing alone (profiles)
        print(json.dumps(timing(2048, 1 << lg)), flush=True)
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[2] == "sweep":            # small batches: the latency route against the throughput route
        for bits in (2048, 1024, 3072, 4096):
            key = synthetic_key(bits, 0x1234567)
            pub = engine.PublicKeyHandle(key.n, bits, key.hs, key.randbits, device=dev)
            for N in (2048, 3072, 4096, 5120, 6144, 7168, 8192, 10240, 12288, 16384):
                g = torch.Generator(device=dev); g.manual_seed(1)
                a = torch.randint(-(2**31), 2**31, (N, pub.ct_words), dtype=torch.int64, device=dev, generator=g).to(torch.int32)
                a[:, -1] &= 0x00FFFFFF
                b = a.flip(0).contiguous()
                out = pub.empty_ct(N)
                row = {"key_bits": bits, "batch": N}
                for name, env in (("default", None), ("throughput", "0"), ("latency", "1000000")):
                    if env is None: os.environ.pop("PAI_LAT_ADD_MAX", None)
                    else: os.environ["PAI_LAT_ADD_MAX"] = env
                    pub.ct_add(a, b, out=out); torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(50): pub.ct_add(a, b, out=out)
                    torch.cuda.synchronize()
                    row[name + "_us"] = round((time.perf_counter() - t0) / 50 * 1e6, 1)
                os.environ.pop("PAI_LAT_ADD_MAX", None)
                print(json.dumps(row), flush=True)
        sys.exit(0)
    for bits in (2048, 1024, 3072, 4096):
        print(json.dumps(check(bits, 4099, rng)), flush=True)
    for bits, B in ((2048, 1 << lg), (1024, 1 << lg), (3072, 1 << (lg - 1)), (4096, 1 << (lg - 2)), (2048, 70000), (2048, 5000)):
        print(json.dumps(timing(bits, B)), flush=True)
