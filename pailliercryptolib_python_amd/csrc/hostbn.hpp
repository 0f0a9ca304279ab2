// Host-side multi-limb integers for key set-up (NOT on the hot path).
//
// The reference derives all key material inside the un-vendored ipcl C++ library
// (ipcl::PublicKey / ipcl::PrivateKey constructors, reached from
// bindings/ipcl_bindings_classes.cpp:17-27,96-101); here the C-ABI library derives the same
// quantities (n^2, hp, hq, p^-1 mod q — SURVEY.md App. D) plus the Montgomery constants of the
// radix-2^29 device representation.  Everything is little-endian u32 limbs, value semantics,
// schoolbook algorithms: sizes are <= 8192 bits and each routine runs once per key.
#pragma once
#include <algorithm>
#include <cstdint>
#include <stdexcept>
#include <vector>

namespace pai {
namespace hbn {

using Limbs = std::vector<uint32_t>;

inline void trim(Limbs& a) { while (!a.empty() && a.back() == 0) a.pop_back(); }
inline Limbs from_u32(const uint32_t* p, size_t n) { Limbs r(p, p + n); trim(r); return r; }
inline Limbs from_u64(uint64_t v) { Limbs r{(uint32_t)v, (uint32_t)(v >> 32)}; trim(r); return r; }
inline bool is_zero(const Limbs& a) { return a.empty(); }
inline bool is_odd(const Limbs& a) { return !a.empty() && (a[0] & 1u); }

inline int bitlen(const Limbs& a) {
    if (a.empty()) return 0;
    return 32 * (int)(a.size() - 1) + (32 - __builtin_clz(a.back()));
}
inline int cmp(const Limbs& a, const Limbs& b) {
    if (a.size() != b.size()) return a.size() < b.size() ? -1 : 1;
    for (size_t i = a.size(); i-- > 0;)
        if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1;
    return 0;
}
inline Limbs add(const Limbs& a, const Limbs& b) {
    Limbs r(std::max(a.size(), b.size()) + 1, 0);
    uint64_t c = 0;
    for (size_t i = 0; i < r.size(); ++i) {
        c += (i < a.size() ? a[i] : 0u);
        c += (i < b.size() ? b[i] : 0u);
        r[i] = (uint32_t)c;
        c >>= 32;
    }
    trim(r);
    return r;
}
// a - b, requires a >= b
inline Limbs sub(const Limbs& a, const Limbs& b) {
    if (cmp(a, b) < 0) throw std::runtime_error("hbn::sub underflow");
    Limbs r(a.size(), 0);
    int64_t c = 0;
    for (size_t i = 0; i < a.size(); ++i) {
        int64_t t = (int64_t)a[i] - (i < b.size() ? b[i] : 0u) + c;
        r[i] = (uint32_t)t;
        c = t >> 32;
    }
    trim(r);
    return r;
}
inline Limbs mul(const Limbs& a, const Limbs& b) {
    if (a.empty() || b.empty()) return {};
    Limbs r(a.size() + b.size(), 0);
    for (size_t i = 0; i < a.size(); ++i) {
        uint64_t c = 0;
        for (size_t j = 0; j < b.size(); ++j) {
            c += (uint64_t)a[i] * b[j] + r[i + j];
            r[i + j] = (uint32_t)c;
            c >>= 32;
        }
        r[i + b.size()] = (uint32_t)c;
    }
    trim(r);
    return r;
}
inline Limbs shl(const Limbs& a, int bits) {
    if (a.empty()) return {};
    int w = bits / 32, s = bits % 32;
    Limbs r(a.size() + w + 1, 0);
    for (size_t i = 0; i < a.size(); ++i) {
        r[i + w] |= a[i] << s;
        if (s) r[i + w + 1] |= a[i] >> (32 - s);
    }
    trim(r);
    return r;
}
inline Limbs shr(const Limbs& a, int bits) {
    int w = bits / 32, s = bits % 32;
    if ((size_t)w >= a.size()) return {};
    Limbs r(a.size() - w, 0);
    for (size_t i = 0; i < r.size(); ++i) {
        r[i] = a[i + w] >> s;
        if (s && i + w + 1 < a.size()) r[i] |= a[i + w + 1] << (32 - s);
    }
    trim(r);
    return r;
}
// low `bits` bits of a
inline Limbs low_bits(const Limbs& a, int bits) {
    Limbs r(a.begin(), a.begin() + std::min(a.size(), (size_t)((bits + 31) / 32)));
    if (bits % 32 && r.size() == (size_t)((bits + 31) / 32)) r.back() &= (1u << (bits % 32)) - 1;
    trim(r);
    return r;
}
// a mod m by binary shift-subtract (a may be much larger than m); O(bits * limbs), set-up only.
inline Limbs mod(const Limbs& a, const Limbs& m) {
    if (m.empty()) throw std::runtime_error("hbn::mod by zero");
    if (cmp(a, m) < 0) return a;
    int sh = bitlen(a) - bitlen(m);
    Limbs r = a;
    for (int s = sh; s >= 0; --s) {
        Limbs t = shl(m, s);
        if (cmp(r, t) >= 0) r = sub(r, t);
    }
    return r;
}
// floor(a / m) by the same loop (exact divisions in validation only)
inline Limbs divq(const Limbs& a, const Limbs& m, Limbs* rem = nullptr) {
    if (m.empty()) throw std::runtime_error("hbn::div by zero");
    Limbs q((a.size() > m.size() ? a.size() - m.size() : 0) + 1, 0), r = a;
    for (int s = bitlen(a) - bitlen(m); s >= 0; --s) {
        Limbs t = shl(m, s);
        if (cmp(r, t) >= 0) { r = sub(r, t); q[s / 32] |= 1u << (s % 32); }
    }
    trim(q);
    if (rem) *rem = r;
    return q;
}
inline Limbs mulmod(const Limbs& a, const Limbs& b, const Limbs& m) { return mod(mul(a, b), m); }

// -m^-1 mod 2^32 for odd m (Newton)
inline uint32_t neg_inv32(uint32_t m0) {
    uint32_t x = m0;                       // correct to 3 bits
    for (int i = 0; i < 5; ++i) x *= 2u - m0 * x;
    return 0u - x;
}

// Word-serial Montgomery arithmetic on the host (radix 2^32), used for modexp during set-up.
struct Mont32 {
    Limbs m;
    size_t L;
    uint32_t m0inv;   // -m^-1 mod 2^32
    Limbs r2;         // R^2 mod m, R = 2^(32 L)
    explicit Mont32(const Limbs& mod_) : m(mod_) {
        if (!is_odd(m)) throw std::runtime_error("Mont32: modulus must be odd");
        L = m.size();
        m0inv = neg_inv32(m[0]);
        Limbs r = mod(shl(Limbs{1u}, 32 * (int)L), m);
        r2 = mulmod(r, r, m);
    }
    Limbs pad(const Limbs& a) const { Limbs r = a; r.resize(L, 0); return r; }
    // a*b*R^-1 mod m, inputs < m (padded or not), output < m trimmed
    Limbs mmul(const Limbs& a_, const Limbs& b_) const {
        Limbs a = pad(a_), b = pad(b_);
        std::vector<uint32_t> t(L + 2, 0);
        for (size_t i = 0; i < L; ++i) {
            uint64_t c = 0;
            for (size_t j = 0; j < L; ++j) {
                c += (uint64_t)a[j] * b[i] + t[j];
                t[j] = (uint32_t)c;
                c >>= 32;
            }
            c += t[L];
            t[L] = (uint32_t)c;
            t[L + 1] = (uint32_t)(c >> 32);
            uint32_t q = t[0] * m0inv;
            c = ((uint64_t)q * m[0] + t[0]) >> 32;
            for (size_t j = 1; j < L; ++j) {
                c += (uint64_t)q * m[j] + t[j];
                t[j - 1] = (uint32_t)c;
                c >>= 32;
            }
            c += t[L];
            t[L - 1] = (uint32_t)c;
            t[L] = t[L + 1] + (uint32_t)(c >> 32);
        }
        Limbs r(t.begin(), t.begin() + L + 1);
        trim(r);
        if (cmp(r, m) >= 0) r = sub(r, m);
        return r;
    }
    Limbs to_mont(const Limbs& a) const { return mmul(a, r2); }
    Limbs from_mont(const Limbs& a) const { return mmul(a, Limbs{1u}); }
    Limbs powmod(const Limbs& base, const Limbs& e) const {
        Limbs x = to_mont(Limbs{1u}), b = to_mont(mod(base, m));
        for (int i = bitlen(e) - 1; i >= 0; --i) {
            x = mmul(x, x);
            if ((e[i / 32] >> (i % 32)) & 1u) x = mmul(x, b);
        }
        return from_mont(x);
    }
};

// a^-1 mod prime p via Fermat (p must be prime; callers verify a*inv == 1 afterwards)
inline Limbs inv_mod_prime(const Limbs& a, const Limbs& p) {
    Mont32 mt(p);
    return mt.powmod(a, sub(p, Limbs{2u}));
}

// a^-1 mod 2^bits for odd a (Hensel/Newton lifting on truncated products)
inline Limbs inv_mod_pow2(const Limbs& a, int bits) {
    if (!is_odd(a)) throw std::runtime_error("inv_mod_pow2: even input");
    Limbs x{0u - neg_inv32(a[0])};          // a^-1 mod 2^32
    for (int prec = 32; prec < bits; prec *= 2) {
        int np = std::min(prec * 2, bits);
        // x = x * (2 - a*x) mod 2^np
        Limbs ax = low_bits(mul(low_bits(a, np), x), np);
        Limbs two = shl(Limbs{1u}, np);     // 2^np
        Limbs t = low_bits(add(sub(two, ax), Limbs{2u}), np);   // (2 - ax) mod 2^np
        x = low_bits(mul(x, t), np);
    }
    return low_bits(x, bits);
}

// ---- radix-2^29 packing used by the device kernels ------------------------------------------
constexpr int RB = 29;
inline std::vector<uint32_t> to_r29(const Limbs& a, int nl) {
    std::vector<uint32_t> r(nl, 0);
    if (bitlen(a) > RB * nl) throw std::runtime_error("to_r29: value does not fit");
    for (int j = 0; j < nl; ++j) {
        int bit = RB * j, w = bit / 32, s = bit % 32;
        uint64_t v = 0;
        if ((size_t)w < a.size()) v = a[w];
        if ((size_t)w + 1 < a.size()) v |= (uint64_t)a[w + 1] << 32;
        r[j] = (uint32_t)(v >> s) & ((1u << RB) - 1);
    }
    return r;
}
inline Limbs from_r29(const uint32_t* r, int nl) {
    Limbs a((size_t)(nl * RB + 31) / 32 + 1, 0);
    for (int j = 0; j < nl; ++j) {
        int bit = RB * j, w = bit / 32, s = bit % 32;
        uint64_t v = (uint64_t)r[j] << s;
        a[w] |= (uint32_t)v;
        a[w + 1] |= (uint32_t)(v >> 32);
    }
    trim(a);
    return a;
}

}  // namespace hbn
}  // namespace pai
