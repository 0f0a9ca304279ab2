// Batch modular inversion modulo n^2 (ct_invert).
// The reference inverts one ciphertext at a time on the CPU with gmpy2.invert
// (ipcl_python.py:272-276, used for negative plaintext multipliers :426-437,470-479).  Bit parity
// needs the true inverse, so it is computed on the device with Montgomery's simultaneous-inversion
// trick arranged as a PRODUCT TREE over halves (paillier_capi.hip: pai_ct_invert): level k+1 holds
// P[i] = a[i] * a[i + h] (h = ceil(count / 2)) — one full-occupancy k_modmul launch per level — until
// at most 64 products are left; those are inverted by k_inv_eea_wave (one WAVE per value), and the way
// down peels two inverses off every node: a[i]^-1 = P[i]^-1 a[i + h], a[i + h]^-1 = P[i]^-1 a[i].
// Cost per element: 3 modular multiplications, all in fully parallel launches, + O(1) extended GCDs per
// call (the first version walked chunks of 32 sequentially per lane group and ran one single-thread
// extended GCD per chunk: 525 ms per 2^20 at 2048-bit keys, 300 ms of it the latency of ONE thread's GCD).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace pai {

// ---- multiword arithmetic on one wave: lane l holds words l*WPL .. l*WPL + WPL - 1 (64*WPL words) -------
// Cross-lane carries are resolved with the ballot trick: G = lanes that generate a carry, P = lanes that
// propagate one (sum all ones / difference all zeros; G and P are disjoint), then the lanes receiving a
// carry are (P + (G << 1)) ^ P and the carry out of the top is bit 64 of that sum or G's top bit.
template <int WPL>
struct WaveInt {
    uint32_t w[WPL];
};

template <int WPL>
__device__ __forceinline__ uint64_t wave_resolve(bool g, bool p, bool& carry_out) {
    const uint64_t G = __ballot(g), P = __ballot(p);
    const uint64_t Gs = G << 1;
    const uint64_t T = P + Gs;
    carry_out = (G >> 63) || (T < P);
    return T ^ P;
}

// d = x - y; returns the borrow out of the top
template <int WPL>
__device__ __forceinline__ bool wave_sub(WaveInt<WPL>& d, const WaveInt<WPL>& x, const WaveInt<WPL>& y) {
    int64_t c = 0;
    uint32_t nz = 0;
#pragma unroll
    for (int k = 0; k < WPL; ++k) {
        c += (int64_t)x.w[k] - (int64_t)y.w[k];
        d.w[k] = (uint32_t)c;
        nz |= d.w[k];
        c >>= 32;
    }
    bool out;
    const uint64_t in = wave_resolve<WPL>(c < 0, nz == 0, out);
    if ((in >> (threadIdx.x & 63)) & 1ull) {
        uint32_t b = 1;
#pragma unroll
        for (int k = 0; k < WPL; ++k) {
            const uint32_t t = d.w[k];
            d.w[k] = t - b;
            b = (b && t == 0) ? 1u : 0u;
        }
    }
    return out;
}

// d = x + y; returns the carry out of the top
template <int WPL>
__device__ __forceinline__ bool wave_add(WaveInt<WPL>& d, const WaveInt<WPL>& x, const WaveInt<WPL>& y) {
    uint64_t c = 0;
    uint32_t ones = 0xFFFFFFFFu;
#pragma unroll
    for (int k = 0; k < WPL; ++k) {
        c += (uint64_t)x.w[k] + y.w[k];
        d.w[k] = (uint32_t)c;
        ones &= d.w[k];
        c >>= 32;
    }
    bool out;
    const uint64_t in = wave_resolve<WPL>(c != 0, ones == 0xFFFFFFFFu, out);
    if ((in >> (threadIdx.x & 63)) & 1ull) {
        uint32_t b = 1;
#pragma unroll
        for (int k = 0; k < WPL; ++k) {
            d.w[k] += b;
            b = (b && d.w[k] == 0) ? 1u : 0u;
        }
    }
    return out;
}

// x >>= 1 with `top` shifted into the most significant bit
template <int WPL>
__device__ __forceinline__ void wave_shr1(WaveInt<WPL>& x, bool top) {
    const int lane = threadIdx.x & 63;
    // lane l <- word 0 of lane l + 1: wave_shl DPP (a VALU move; __shfl_down goes through the LDS pipe, ~100 cycles
    // of latency on the critical path of every halving)
    uint32_t next = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x.w[0], 0x130, 0xF, 0xF, true);
    if (lane == 63) next = top ? 1u : 0u;
#pragma unroll
    for (int k = 0; k < WPL; ++k) {
        const uint32_t hi = (k + 1 < WPL) ? x.w[k + 1] : next;
        x.w[k] = (x.w[k] >> 1) | (hi << 31);
    }
}

template <int WPL>
__device__ __forceinline__ bool wave_is_zero(const WaveInt<WPL>& x) {
    uint32_t nz = 0;
#pragma unroll
    for (int k = 0; k < WPL; ++k) nz |= x.w[k];
    return __ballot(nz != 0) == 0;
}

// One wave per value: out = a^-1 mod M (M odd), `words` 32-bit words each (words <= 64 WPL).
// Invariants: u = x1 a, v = x2 a (mod M), v odd, x1, x2 in [0, M).  An odd u is replaced by |u - v| (the smaller
// of the two stays in v, the x's follow) which is even; an even u is halved together with x1 (mod M).  u reaches
// 0 after at most 2 * bits + 2 halvings, leaving v = gcd(a, M) and, when that is 1, x2 = a^-1.  Every decision is
// wave-uniform.  *fail is set when some gcd is not 1.
template <int WPL>
__global__ void __launch_bounds__(64)
k_inv_eea_wave(const uint32_t* __restrict__ mod, const uint32_t* __restrict__ a_in, uint32_t* __restrict__ out, int words,
               int max_steps, int* __restrict__ fail) {
    const int lane = threadIdx.x & 63;
    const size_t row = (size_t)blockIdx.x * words;
    WaveInt<WPL> u, v, x1, x2, m;
#pragma unroll
    for (int k = 0; k < WPL; ++k) {
        const int j = lane * WPL + k;
        u.w[k] = j < words ? a_in[row + j] : 0u;
        m.w[k] = j < words ? mod[j] : 0u;
        v.w[k] = m.w[k];
        x1.w[k] = 0;
        x2.w[k] = 0;
    }
    if (lane == 0) x1.w[0] = 1;
    for (int step = 0; step < max_steps; ++step) {
        // once u is 0 nothing below touches v or x2 (u stays even: only u and x1 are halved), so the exit test can be
        // sparse; lane 0's low words travel through SGPRs (readfirstlane), not the LDS pipe
        if ((step & 15) == 0 && wave_is_zero<WPL>(u)) break;
        const bool u_odd = (uint32_t)__builtin_amdgcn_readfirstlane((int)u.w[0]) & 1u;
        if (u_odd) {
            WaveInt<WPL> d;
            const bool lt = wave_sub<WPL>(d, u, v);
            if (lt) {                     // u < v: (u, v) <- (v - u, u), the coefficients swap roles
                wave_sub<WPL>(d, v, u);
                v = u;
                const WaveInt<WPL> t = x1;
                x1 = x2;
                x2 = t;
            }
            u = d;
            WaveInt<WPL> e;
            if (wave_sub<WPL>(e, x1, x2)) wave_add<WPL>(e, e, m);
            x1 = e;
        }
        // u is even: halve it, and x1 modulo M
        wave_shr1<WPL>(u, false);
        bool top = false;
        if ((uint32_t)__builtin_amdgcn_readfirstlane((int)x1.w[0]) & 1u) top = wave_add<WPL>(x1, x1, m);
        wave_shr1<WPL>(x1, top);
    }
    // v = gcd(a, M)
    uint32_t diff = 0;
#pragma unroll
    for (int k = 0; k < WPL; ++k) diff |= v.w[k] ^ ((lane == 0 && k == 0) ? 1u : 0u);
    const bool bad = __ballot(diff != 0) != 0 || !wave_is_zero<WPL>(u);
    if (bad && lane == 0) atomicExch(fail, 1);
#pragma unroll
    for (int k = 0; k < WPL; ++k) {
        const int j = lane * WPL + k;
        if (j < words) out[row + j] = x2.w[k];
    }
}

}  // namespace pai
