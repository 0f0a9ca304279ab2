// Host-side key generation for the C ABI (pai_keygen, pai_host_modexp): the native counterpart of
// ipcl::generateKeypair (reference bindings/ipcl_bindings.cpp:12-15, timed by bench/bench_ipcl_python.py:13-19 BM_KeyGen).
// Not a hot operation (one call per key): radix-2^64 Montgomery arithmetic on the host cores, fixed 4-bit windows,
// incremental prime search over a small-prime sieve with Miller-Rabin, the survivors of the sieve tested on up to 16 threads.
// Randomness: getrandom(2) (the kernel CSPRNG) unless the caller supplies a seed (tests: reproducible keys through
// a ChaCha-free splitmix stream — NOT for production keys, and the ABI says so).
#pragma once
#include <sys/random.h>

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstring>
#include <string.h>
#include <stdexcept>
#include <thread>
#include <vector>

namespace pai {
namespace kg {

typedef unsigned __int128 u128;
constexpr int MAXL = 130;                 // 64-bit limbs: moduli up to 8320 bits (n^2 of a 4096-bit key)

struct Rng {
    bool seeded = false;
    uint64_t s = 0;
    explicit Rng(const uint64_t* seed) {
        if (seed) { seeded = true; s = *seed; }
    }
    void fill(uint64_t* out, int n) {
        if (seeded) {
            for (int i = 0; i < n; ++i) {                       // splitmix64
                uint64_t z = (s += 0x9E3779B97F4A7C15ull);
                z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
                z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
                out[i] = z ^ (z >> 31);
            }
            return;
        }
        size_t need = (size_t)n * 8, got = 0;
        while (got < need) {
            ssize_t k = getrandom((char*)out + got, need - got, 0);
            if (k <= 0) throw std::runtime_error("getrandom failed");
            got += (size_t)k;
        }
    }
};

inline int bitlen(const uint64_t* a, int L) {
    for (int i = L - 1; i >= 0; --i)
        if (a[i]) return 64 * i + 64 - __builtin_clzll(a[i]);
    return 0;
}
inline int cmp(const uint64_t* a, const uint64_t* b, int L) {
    for (int i = L - 1; i >= 0; --i)
        if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1;
    return 0;
}
inline uint64_t sub_n(uint64_t* r, const uint64_t* a, const uint64_t* b, int L) {
    uint64_t borrow = 0;
    for (int i = 0; i < L; ++i) {
        u128 t = (u128)a[i] - b[i] - borrow;
        r[i] = (uint64_t)t;
        borrow = (uint64_t)(t >> 64) & 1u;
    }
    return borrow;
}
inline uint64_t mod_small(const uint64_t* a, int L, uint32_t d) {
    uint64_t r = 0;
    for (int i = L - 1; i >= 0; --i) {
        u128 t = ((u128)r << 64) | a[i];
        r = (uint64_t)(t % d);
    }
    return r;
}

// Montgomery context for an odd modulus of L 64-bit limbs
struct Mont {
    int L = 0;
    uint64_t m[MAXL], r1[MAXL], r2[MAXL], m0inv = 0;          // r1 = R mod m, r2 = R^2 mod m
    Mont(const uint64_t* mod, int L_) : L(L_) {
        if (L < 1 || L > MAXL || !(mod[0] & 1)) throw std::runtime_error("Mont: odd modulus of 1..130 limbs expected");
        std::memcpy(m, mod, 8 * L);
        uint64_t x = m[0];
        for (int i = 0; i < 6; ++i) x *= 2 - m[0] * x;
        m0inv = 0 - x;
        // R mod m by 64 L doublings of (2^(bitlen-1) mod m ... ) — simple: start from 1, double 64 L times with reduction
        uint64_t t[MAXL] = {0};
        t[0] = 1;
        auto dbl = [&](uint64_t* v) {
            uint64_t c = 0;
            for (int i = 0; i < L; ++i) { uint64_t n = (v[i] << 1) | c; c = v[i] >> 63; v[i] = n; }
            if (c || cmp(v, m, L) >= 0) sub_n(v, v, m, L);
        };
        for (int i = 0; i < 64 * L; ++i) dbl(t);
        std::memcpy(r1, t, 8 * L);
        for (int i = 0; i < 64 * L; ++i) dbl(t);
        std::memcpy(r2, t, 8 * L);
    }
    // out = a b R^-1 mod m (operands < m)
    void mul(uint64_t* out, const uint64_t* a, const uint64_t* b) const {
        uint64_t t[MAXL + 2];
        std::memset(t, 0, 8 * (L + 2));
        for (int i = 0; i < L; ++i) {
            u128 c = 0;
            const uint64_t bi = b[i];
            for (int j = 0; j < L; ++j) {
                c += (u128)a[j] * bi + t[j];
                t[j] = (uint64_t)c;
                c >>= 64;
            }
            c += t[L];
            t[L] = (uint64_t)c;
            t[L + 1] = (uint64_t)(c >> 64);
            const uint64_t q = t[0] * m0inv;
            c = ((u128)q * m[0] + t[0]) >> 64;
            for (int j = 1; j < L; ++j) {
                c += (u128)q * m[j] + t[j];
                t[j - 1] = (uint64_t)c;
                c >>= 64;
            }
            c += t[L];
            t[L - 1] = (uint64_t)c;
            t[L] = t[L + 1] + (uint64_t)(c >> 64);
        }
        if (t[L] || cmp(t, m, L) >= 0) sub_n(t, t, m, L);
        std::memcpy(out, t, 8 * L);
    }
    // out = base^e mod m, plain in / plain out (base < m); fixed 4-bit windows
    void pow(uint64_t* out, const uint64_t* base, const uint64_t* e, int eL) const {
        std::vector<uint64_t> tab((size_t)16 * L);
        std::memcpy(&tab[0], r1, 8 * L);
        mul(&tab[L], base, r2);
        for (int k = 2; k < 16; ++k) mul(&tab[(size_t)k * L], &tab[(size_t)(k - 1) * L], &tab[L]);
        uint64_t x[MAXL];
        std::memcpy(x, r1, 8 * L);
        const int eb = bitlen(e, eL);
        for (int w = (eb + 3) / 4 - 1; w >= 0; --w) {
            for (int s = 0; s < 4; ++s) mul(x, x, x);
            const int d = (int)((e[(4 * w) / 64] >> ((4 * w) % 64)) & 15u);
            if (d) mul(x, x, &tab[(size_t)d * L]);
        }
        uint64_t one[MAXL] = {0};
        one[0] = 1;
        mul(out, x, one);
    }
};

inline const std::vector<uint32_t>& small_primes() {
    static const std::vector<uint32_t> P = [] {
        const int LIM = 1 << 16;
        std::vector<char> comp(LIM, 0);
        std::vector<uint32_t> p;
        for (int i = 3; i < LIM; i += 2) {
            if (comp[i]) continue;
            p.push_back((uint32_t)i);
            for (long j = (long)i * i; j < LIM; j += 2 * i) comp[j] = 1;
        }
        return p;
    }();
    return P;
}

// Miller-Rabin: `rounds` random bases after base 2 (n odd, > 3)
inline bool miller_rabin(const uint64_t* n, int L, int rounds, Rng& rng, bool base2 = true) {
    Mont mt(n, L);
    uint64_t nm1[MAXL], d[MAXL], mone[MAXL];
    std::memcpy(nm1, n, 8 * L);
    nm1[0] &= ~1ull;
    std::memcpy(d, nm1, 8 * L);
    int r = 0;
    while (!(d[0] & 1)) {                                   // d = (n-1) / 2^r
        for (int i = 0; i < L; ++i) d[i] = (d[i] >> 1) | (i + 1 < L ? d[i + 1] << 63 : 0);
        ++r;
    }
    sub_n(mone, mt.m, mt.r1, L);                            // -1 in Montgomery form
    const int nb = bitlen(n, L);
    for (int it = base2 ? 0 : 1; it <= rounds; ++it) {
        uint64_t a[MAXL];
        if (it == 0) {
            std::memset(a, 0, 8 * L);
            a[0] = 2;
        } else {
            do {
                rng.fill(a, L);
                const int top = (nb - 1) % 64;              // a < 2^(nb-1) <= n
                a[L - 1] &= top ? ((1ull << top) - 1) : 0;
            } while (bitlen(a, L) < 2);
        }
        // x = a^d in Montgomery form
        uint64_t x[MAXL], am[MAXL];
        mt.mul(am, a, mt.r2);
        std::memcpy(x, mt.r1, 8 * L);
        // 4-bit windows
        std::vector<uint64_t> tab((size_t)16 * L);
        std::memcpy(&tab[0], mt.r1, 8 * L);
        std::memcpy(&tab[L], am, 8 * L);
        for (int k = 2; k < 16; ++k) mt.mul(&tab[(size_t)k * L], &tab[(size_t)(k - 1) * L], am);
        const int db = bitlen(d, L);
        for (int w = (db + 3) / 4 - 1; w >= 0; --w) {
            for (int s = 0; s < 4; ++s) mt.mul(x, x, x);
            const int dg = (int)((d[(4 * w) / 64] >> ((4 * w) % 64)) & 15u);
            if (dg) mt.mul(x, x, &tab[(size_t)dg * L]);
        }
        if (cmp(x, mt.r1, L) == 0 || cmp(x, mone, L) == 0) continue;
        bool witness = true;
        for (int s = 1; s < r; ++s) {
            mt.mul(x, x, x);
            if (cmp(x, mone, L) == 0) { witness = false; break; }
            if (cmp(x, mt.r1, L) == 0) break;
        }
        if (witness) return false;
    }
    return true;
}

// Worker threads of a prime search (the two primes of a key are searched one after the other)
inline int search_threads() {
    const unsigned hw = std::thread::hardware_concurrency();
    return (int)std::max(1u, std::min(16u, hw));
}

// A random prime of exactly `bits` bits with the two top bits set (so a product of two has exactly 2 bits bits);
// mod4_3: congruent to 3 modulo 4.  Incremental search from a random start over a sieve of the odd primes < 2^16: the
// result is the FIRST prime at or after the start in steps of 2 (4), whatever the number of threads — the sieve's
// survivors are handed out in order to `threads` workers (base-2 Miller-Rabin each), the smallest index that passes wins,
// and its `rounds` random-base rounds are split over the same workers.  Seeded searches are reproducible: the bases come
// from generators derived from the caller's, and a prime is a prime under any bases.
inline void random_prime(uint64_t* out, int bits, bool mod4_3, int rounds, Rng& rng, int threads = 0,
                         const std::vector<uint32_t>* not_one_mod = nullptr) {
    const int L = (bits + 63) / 64;
    const auto& sp = small_primes();
    const uint32_t step = mod4_3 ? 4 : 2;
    const int WIN = 4096;                                   // candidates per sieve window
    const int T = threads > 0 ? threads : search_threads();
    for (;;) {
        uint64_t c[MAXL];
        rng.fill(c, L);
        uint64_t base_seed = 0;
        rng.fill(&base_seed, 1);
        const int top = bits % 64;
        if (top) c[L - 1] &= (1ull << top) - 1;
        const int hb = (bits - 1) % 64, hb2 = (bits - 2) % 64;
        c[(bits - 1) / 64] |= 1ull << hb;
        c[(bits - 2) / 64] |= 1ull << hb2;
        c[0] |= mod4_3 ? 3 : 1;
        std::vector<char> comp(WIN, 0);
        for (uint32_t p : sp) {
            const uint64_t r = mod_small(c, L, p);
            // smallest k >= 0 with r + k step == 0 (mod p)
            uint64_t inv_step = 0;
            {   // step^-1 mod p (p odd prime, step in {2, 4})
                const uint64_t half = (p + 1) / 2;
                inv_step = step == 2 ? half : (half * half) % p;
            }
            uint64_t k = ((p - r) % p) * inv_step % p;
            for (; k < (uint64_t)WIN; k += p) comp[k] = 1;
        }
        if (not_one_mod) {
            // ... and no candidate congruent to 1 modulo these primes (the small odd prime factors of the other prime's
            // p - 1: gcd(p - 1, q - 1) = 2 then only fails on a common LARGE factor)
            for (uint32_t l : *not_one_mod) {
                const uint64_t r = mod_small(c, L, l), half = (l + 1) / 2;
                const uint64_t inv_step = step == 2 ? half : (half * half) % l;
                uint64_t k = ((l + 1 - r) % l) * inv_step % l;
                for (; k < (uint64_t)WIN; k += l) comp[k] = 1;
            }
        }
        std::vector<int> ks;                                // the survivors, in order
        for (int k = 0; k < WIN; ++k) if (!comp[k]) ks.push_back(k);
        // candidate number k; false when it runs over the top (every later one does too)
        auto candidate = [&](int k, uint64_t* cand) -> bool {
            u128 carry = (u128)k * step;
            for (int i = 0; i < L; ++i) {
                carry += c[i];
                cand[i] = (uint64_t)carry;
                carry >>= 64;
            }
            return !carry && bitlen(cand, L) == bits;
        };
        auto run = [&](int n, auto&& body) {                // body(t) on n threads, exceptions re-thrown here
            std::vector<std::thread> th;
            std::vector<std::exception_ptr> err((size_t)n);
            for (int t = 1; t < n; ++t) th.emplace_back([&, t] { try { body(t); } catch (...) { err[(size_t)t] = std::current_exception(); } });
            try { body(0); } catch (...) { err[0] = std::current_exception(); }
            for (auto& x : th) x.join();
            for (auto& e : err) if (e) std::rethrow_exception(e);
        };
        int from = 0;
        const int nk = (int)ks.size();
        while (from < nk) {
            // smallest index >= from whose candidate passes base 2
            std::atomic<int> next(from), best(nk);
            run(T, [&](int) {
                Rng none(&base_seed);                       // (base 2 draws nothing)
                for (;;) {
                    const int i = next.fetch_add(1, std::memory_order_relaxed);
                    if (i >= nk || i > best.load(std::memory_order_relaxed)) return;
                    uint64_t cand[MAXL];
                    const bool in_range = candidate(ks[(size_t)i], cand);
                    const bool pass = in_range && miller_rabin(cand, L, 0, none);
                    explicit_bzero(cand, sizeof(cand));
                    if (!in_range) return;                   // ran over the top: nothing beyond
                    if (pass) {
                        int cur = best.load(std::memory_order_relaxed);
                        while (i < cur && !best.compare_exchange_weak(cur, i, std::memory_order_relaxed)) {}
                    }
                }
            });
            const int hit = best.load();
            if (hit >= nk) break;                            // nothing in this window: new start
            uint64_t cand[MAXL];
            candidate(ks[(size_t)hit], cand);
            // `rounds` random bases, split over the workers
            std::atomic<bool> composite(false);
            const int VT = std::max(1, std::min(T, rounds));
            run(VT, [&](int t) {
                const int mine = rounds / VT + (t < rounds % VT ? 1 : 0);
                if (mine == 0) return;
                uint64_t sd = base_seed + 0xD1B54A32D192ED03ull * (uint64_t)(t + 1) + (uint64_t)hit;
                Rng r(rng.seeded ? &sd : nullptr);
                if (!miller_rabin(cand, L, mine, r, false)) composite.store(true);
            });
            if (!composite.load()) {
                std::memcpy(out, cand, 8 * L);
                explicit_bzero(cand, sizeof(cand));             // the prime and its search start leave this stack frame
                explicit_bzero(c, sizeof(c));
                return;
            }
            from = hit + 1;                                  // a base-2 pseudoprime (never seen): carry on behind it
        }
    }
}

// gcd(a, b) == 2 ?  (binary GCD on copies)
inline bool gcd_is_two(const uint64_t* a_, const uint64_t* b_, int L) {
    uint64_t a[MAXL], b[MAXL];
    std::memcpy(a, a_, 8 * L);
    std::memcpy(b, b_, 8 * L);
    auto shr1 = [&](uint64_t* v) { for (int i = 0; i < L; ++i) v[i] = (v[i] >> 1) | (i + 1 < L ? v[i + 1] << 63 : 0); };
    auto zero = [&](const uint64_t* v) { for (int i = 0; i < L; ++i) if (v[i]) return false; return true; };
    int shift = 0;
    while (!((a[0] | b[0]) & 1)) { shr1(a); shr1(b); ++shift; }
    while (!(a[0] & 1)) shr1(a);
    while (!zero(b)) {
        while (!(b[0] & 1)) shr1(b);
        if (cmp(a, b, L) > 0) std::swap_ranges(a, a + L, b);
        sub_n(b, b, a, L);
    }
    // gcd = a << shift
    bool two = shift == 1 && a[0] == 1;
    for (int i = 1; i < L && two; ++i) if (a[i]) two = false;
    explicit_bzero(a, sizeof(a));                               // copies of p - 1, q - 1
    explicit_bzero(b, sizeof(b));
    return two;
}

// Two primes for a key of n_bits bits (n_bits a multiple of 128): p != q, p q of exactly n_bits bits; DJN keys
// (upstream's constraint): p = q = 3 (mod 4), gcd(p-1, q-1) = 2.  gcd(n, phi) = 1 follows from
// equal prime sizes.  Outputs: n_bits/128 64-bit limbs each.
inline void generate_primes(int n_bits, bool djn, const uint64_t* seed, uint64_t* p, uint64_t* q) {
    const int half = n_bits / 2, L = (half + 63) / 64;
    const int rounds = 24;
    uint64_t s1 = seed ? *seed * 2 : 0;
    Rng r1(seed ? &s1 : nullptr);
    random_prime(p, half, djn, rounds, r1);
    std::vector<uint32_t> avoid;
    if (djn) {
        for (uint32_t l : small_primes()) {
            if (l > 2 && mod_small(p, L, l) == 1) avoid.push_back(l);
        }
    }
    for (int attempt = 0;; ++attempt) {
        uint64_t s2 = seed ? *seed * 2 + 1 + 0x1000ull * attempt : 0;
        Rng r2(seed ? &s2 : nullptr);
        random_prime(q, half, djn, rounds, r2, 0, djn ? &avoid : nullptr);
        if (cmp(p, q, L) == 0) continue;
        if (djn) {
            uint64_t pm1[MAXL], qm1[MAXL];
            std::memcpy(pm1, p, 8 * L); pm1[0] &= ~1ull;
            std::memcpy(qm1, q, 8 * L); qm1[0] &= ~1ull;
            const bool ok = gcd_is_two(pm1, qm1, L);
            explicit_bzero(pm1, sizeof(pm1));
            explicit_bzero(qm1, sizeof(qm1));
            if (!ok) continue;
        }
        return;
    }
}

}  // namespace kg
}  // namespace pai
