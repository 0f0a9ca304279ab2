// Batched modular inversion of ciphertexts: out_i = ct_i^-1 mod n^2.
//
// The reference inverts one ciphertext at a time on the CPU with gmpy2.invert
// (ipcl_python.py:272-276, used for negative plaintext multipliers :426-437,470-479).  Bit parity
// needs the true inverse, so it is computed on the device with Montgomery's simultaneous-inversion
// trick: every lane group walks a chunk of K consecutive ciphertexts and keeps the running products
// (k_inv_prefix), one thread per chunk inverts the chunk product with a branch-light binary extended
// GCD on 32-bit words (k_inv_eea), and a second walk in reverse order peels the individual inverses
// off (k_inv_back).  Cost per element: 6 Montgomery multiplications + 1/K of an extended GCD.
#pragma once
#include "kernels_common.hpp"

namespace pai {

// prefix[i] (raw radix-29 limbs, Montgomery form) = (c_first ... c_i) R ; tot[chunk] = plain product
template <class G>
__global__ void __launch_bounds__(BLOCK_THREADS, 2)
k_inv_prefix(const MontCtx* __restrict__ ctx, const uint32_t* __restrict__ ct, int w32, int n, int K,
             uint32_t* __restrict__ prefix, uint32_t* __restrict__ tot) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    typename G::NM nm;
    load_modulus<G>(nm, ctx, lds);
    const uint32_t n0inv = ctx->n0inv;
    const int t = G::gl();
    const int nchunks = (n + K - 1) / K;
    const int tiles = (nchunks + G::EPB - 1) / G::EPB;
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int ch = tile * G::EPB + G::elem();
        const bool live = ch < nchunks;
        const int chs = live ? ch : nchunks - 1;
        const int first = chs * K;
        const int last = min(first + K, n) - 1;
        uint32_t r2[G::NLL], p[G::NLL];
        load_const_slice<G>(r2, ctx->r2);
        load_const_slice<G>(p, ctx->one);
#pragma unroll 1
        for (int k = 0; k < K; ++k) {
            const int i = min(first + k, last);            // chunks shorter than K repeat their last element (not stored)
            const bool store = live && (first + k <= last);
            uint32_t c[G::NLL];
            load_elem<G>(c, ct + (size_t)i * w32, w32);
            mm_times<G>(c, r2, lds, nm, n0inv);            // c R
            if (first + k > last) load_const_slice<G>(c, ctx->one);   // past the end of a short chunk: multiply by 1
            mm_times<G>(p, c, lds, nm, n0inv);
            if (store) {
#pragma unroll
                for (int j = 0; j < G::NLL; ++j) prefix[(size_t)i * G::NL + G::NLL * t + j] = p[j];
            }
        }
        uint32_t one[G::NLL];
        set_plain_one<G>(one);
        mm_times<G>(p, one, lds, nm, n0inv);
        cond_sub<G::NLL, G::T>(p, nm);
        if (live) store_elem<G>(p, tot + (size_t)ch * w32, w32, lds);
    }
}

// out_i = ct_i^-1 from the prefix products and the inverted chunk products
template <class G>
__global__ void __launch_bounds__(BLOCK_THREADS, 2)
k_inv_back(const MontCtx* __restrict__ ctx, const uint32_t* __restrict__ ct, int w32, int n, int K,
           const uint32_t* __restrict__ prefix, const uint32_t* __restrict__ tot_inv, uint32_t* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    typename G::NM nm;
    load_modulus<G>(nm, ctx, lds);
    const uint32_t n0inv = ctx->n0inv;
    const int t = G::gl();
    const int nchunks = (n + K - 1) / K;
    const int tiles = (nchunks + G::EPB - 1) / G::EPB;
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int ch = tile * G::EPB + G::elem();
        const bool live = ch < nchunks;
        const int chs = live ? ch : nchunks - 1;
        const int first = chs * K;
        const int last = min(first + K, n) - 1;
        const int len = last - first + 1;
        uint32_t inv[G::NLL];                              // (c_first .. c_k)^-1 R, k running down
        {
            uint32_t r2[G::NLL];
            load_elem<G>(inv, tot_inv + (size_t)chs * w32, w32);
            load_const_slice<G>(r2, ctx->r2);
            mm_times<G>(inv, r2, lds, nm, n0inv);
        }
#pragma unroll 1
        for (int kk = K - 1; kk >= 0; --kk) {
            const bool act = kk < len;                     // lanes of shorter chunks idle through the first rounds
            const int i = first + (act ? kk : len - 1);
            uint32_t x[G::NLL];
#pragma unroll
            for (int j = 0; j < G::NLL; ++j) x[j] = inv[j];
            {
                uint32_t pp[G::NLL];
                const bool has_prev = act && i > first;
                const uint32_t* src = has_prev ? prefix + (size_t)(i - 1) * G::NL : ctx->one;   // one: x * 1
#pragma unroll
                for (int j = 0; j < G::NLL; ++j) pp[j] = src[G::NLL * t + j];
                mm_times<G>(x, pp, lds, nm, n0inv);        // c_i^-1 R
            }
            {   // inv <- inv * c_i  (only while more elements remain below)
                uint32_t c[G::NLL], r2[G::NLL];
                load_elem<G>(c, ct + (size_t)i * w32, w32);
                load_const_slice<G>(r2, ctx->r2);
                mm_times<G>(c, r2, lds, nm, n0inv);
                uint32_t nxt[G::NLL];
#pragma unroll
                for (int j = 0; j < G::NLL; ++j) nxt[j] = inv[j];
                mm_times<G>(nxt, c, lds, nm, n0inv);
#pragma unroll
                for (int j = 0; j < G::NLL; ++j) inv[j] = act ? nxt[j] : inv[j];
            }
            uint32_t one[G::NLL];
            set_plain_one<G>(one);
            mm_times<G>(x, one, lds, nm, n0inv);
            cond_sub<G::NLL, G::T>(x, nm);
            if (live && act) store_elem<G>(x, out + (size_t)i * w32, w32, lds);
        }
    }
}

// One thread per value: x = a^-1 mod M (M odd, gcd(a, M) = 1), W 32-bit words each.
// Invariants: u = x1 a, v = x2 a (mod M), v odd.  Each step makes u even (|u - v|, keeping the smaller
// of the two in v) and halves it; u reaches 0 after at most 2*bits steps, leaving x2 = a^-1.
template <int W>
__global__ void __launch_bounds__(64)
k_inv_eea(const uint32_t* __restrict__ mod, const uint32_t* __restrict__ a_in, uint32_t* __restrict__ out, int count,
          int max_steps, int* __restrict__ fail) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < count;
    const int is = live ? i : count - 1;
    uint32_t u[W], v[W], x1[W], x2[W];
#pragma unroll 4
    for (int k = 0; k < W; ++k) { u[k] = a_in[(size_t)is * W + k]; v[k] = mod[k]; x1[k] = 0; x2[k] = 0; }
    x1[0] = 1;
    bool done = false;
    for (int step = 0; step < max_steps; ++step) {
        if ((step & 31) == 0) {
            uint32_t nz = 0;
#pragma unroll 4
            for (int k = 0; k < W; ++k) nz |= u[k];
            done = (nz == 0);
            if (__all(done)) break;
        }
        if (u[0] & 1u) {
            // d = u - v ; if negative: v <- u, d <- -d, x's swap roles
            uint64_t borrow = 0;
            uint32_t lt = 0;
            {
                int64_t c = 0;
#pragma unroll 4
                for (int k = 0; k < W; ++k) { c += (int64_t)u[k] - (int64_t)v[k]; c >>= 32; }
                lt = (c < 0) ? 1u : 0u;
            }
            (void)borrow;
            int64_t c = 0, cx = 0;
#pragma unroll 4
            for (int k = 0; k < W; ++k) {
                const uint32_t uu = u[k], vv = v[k], a1 = x1[k], a2 = x2[k];
                const uint32_t big = lt ? vv : uu, small = lt ? uu : vv;
                const uint32_t xb = lt ? a2 : a1, xs = lt ? a1 : a2;
                c += (int64_t)big - (int64_t)small;
                u[k] = (uint32_t)c;
                c >>= 32;
                v[k] = small;
                cx += (int64_t)xb - (int64_t)xs;
                x1[k] = (uint32_t)cx;
                cx >>= 32;
                x2[k] = xs;
            }
            if (cx < 0) {                                   // x1 += M
                uint64_t cc = 0;
#pragma unroll 4
                for (int k = 0; k < W; ++k) { cc += (uint64_t)x1[k] + mod[k]; x1[k] = (uint32_t)cc; cc >>= 32; }
            }
        }
        // u >>= 1 ; x1 <- x1 / 2 mod M
        const uint32_t odd = x1[0] & 1u;
        uint64_t cc = 0;
        uint32_t prev_x = 0, prev_u = 0;
#pragma unroll 4
        for (int k = 0; k < W; ++k) {
            cc += (uint64_t)x1[k] + (odd ? mod[k] : 0u);
            const uint32_t xs = (uint32_t)cc;
            cc >>= 32;
            if (k > 0) { x1[k - 1] = (prev_x >> 1) | (xs << 31); u[k - 1] = (prev_u >> 1) | (u[k] << 31); }
            prev_x = xs;
            prev_u = u[k];
        }
        x1[W - 1] = (prev_x >> 1) | ((uint32_t)cc << 31);
        u[W - 1] = prev_u >> 1;
    }
    // success iff u == 0 and v == 1
    uint32_t nz = 0, v1 = v[0] ^ 1u;
#pragma unroll 4
    for (int k = 0; k < W; ++k) { nz |= u[k]; if (k) v1 |= v[k]; }
    if (live) {
        if (nz != 0 || v1 != 0) atomicAdd(fail, 1);
#pragma unroll 4
        for (int k = 0; k < W; ++k) out[(size_t)i * W + k] = x2[k];
    }
}

}  // namespace pai
