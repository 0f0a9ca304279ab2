// Generic batched modular kernels on an arbitrary odd modulus (MontCtx):
//   k_modmul        out_i = a_i * b_i mod M                  (ct+ct add: CipherText::operator+, classes.cpp:318-321)
//   k_modexp_fixed  out_i = base_i ^ E mod M, E wave-uniform  (CRT-decrypt halves, standard obfuscator r^n)
//   k_modexp_var    out_i = base_i ^ e_i mod M                (ct*pt: CipherText::operator*, classes.cpp:324-325)
// All operands are packed little-endian u32 words, row-major [N][W32]; results are canonical residues.
#pragma once
#include "kernels_common.hpp"

namespace pai {

// ---------------------------------------------------------------------------------------------
// out_i = a_i * b_i mod M on canonical packed rows.  Tiles move HBM <-> LDS with coalesced full-width copies
// (load_tile / store_tile); the arithmetic is one or two Montgomery products per element:
//   mode MODMUL_FULL   out = a*b mod M          (a*b*R^-1, then * R^2 * R^-1; b_bcast: b*R is formed once per block,
//                                               so a broadcast addend costs ONE product per element)
//   mode MODMUL_MONT   out = a*b*R^-1 mod M     (canonical residue of the Montgomery product: the body of the
//                                               product trees of pai_ct_invert / pai_ct_prod, which keep track of
//                                               the power of R per tree level on the host)
template <class G>
__global__ void __launch_bounds__(BLOCK_THREADS, 2)
k_modmul(const MontCtx* __restrict__ ctx, const uint32_t* a, const uint32_t* b, uint32_t* out, int n, int w32, int b_bcast, int mode) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    uint32_t* stage = lds + G::LDS_WORDS + G::NL;        // behind the operand buffer and the modulus copy
    typename G::NM nm;
    load_modulus<G>(nm, ctx, lds);
    const uint32_t n0inv = ctx->n0inv;
    const int tiles = (n + G::EPB - 1) / G::EPB;
    uint32_t y[G::NLL];
    if (b_bcast) {                                       // the shared operand, once per block
        load_tile<G>(stage, b, 1, w32, true);
        __syncthreads();
        unpack_row<G>(y, stage);
        if (mode == MODMUL_FULL) {
            uint32_t r2[G::NLL];
            load_const_slice<G>(r2, ctx->r2);
            mm_times<G>(y, r2, lds, nm, n0inv);          // b*R (lazy, < 2M)
        }
        __syncthreads();
    }
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int row0 = tile * G::EPB;
        const int rows = min(G::EPB, n - row0);
        uint32_t x[G::NLL];
        load_tile<G>(stage, a + (size_t)row0 * w32, rows, w32);
        __syncthreads();
        unpack_row<G>(x, stage);
        if (!b_bcast) {
            __syncthreads();
            load_tile<G>(stage, b + (size_t)row0 * w32, rows, w32);
            __syncthreads();
            unpack_row<G>(y, stage);
        }
        mm_times<G>(x, y, lds, nm, n0inv);               // a*b*R^-1 (a*b when y = b*R)
        if (mode == MODMUL_FULL && !b_bcast) {
            uint32_t r2[G::NLL];
            load_const_slice<G>(r2, ctx->r2);
            mm_times<G>(x, r2, lds, nm, n0inv);
        }
        cond_sub<G::NLL, G::T>(x, nm);
        __syncthreads();                                 // every lane has unpacked its operands
        pack_row<G>(x, stage, w32, lds);
        __syncthreads();
        store_tile<G>(stage, out + (size_t)row0 * w32, rows, w32);
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// Fixed-window (W bits) exponentiation with a wave-uniform exponent.  The 2^W-entry table of each
// resident element lives in a global scratch area laid out [entry][limb][slot] so that a wave reads
// one entry with coalesced 128/256-byte rows; slot = blockIdx.x*EPB + element.
template <class G, int W>
__global__ void __launch_bounds__(BLOCK_THREADS, 2)
k_modexp_fixed(const MontCtx* __restrict__ ctx, const uint32_t* __restrict__ base, int base_w32,
               const uint32_t* __restrict__ expo, int ewords, int ebits,
               uint32_t* __restrict__ out, int out_w32, int n, uint32_t* __restrict__ table, int keep_mont) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int t = G::gl();
    typename G::NM nm;
    load_modulus<G>(nm, ctx, lds);
    const uint32_t n0inv = ctx->n0inv;
    const size_t nslots = (size_t)gridDim.x * G::EPB;
    const size_t slot = (size_t)blockIdx.x * G::EPB + G::elem();
    auto tbl = [&](int entry, int j) -> uint32_t& {
        return table[((size_t)entry * G::NL + (G::NLL * t + j)) * nslots + slot];
    };
    const int nwin = (ebits + W - 1) / W;
    const int tiles = (n + G::EPB - 1) / G::EPB;
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int ei = tile * G::EPB + G::elem();
        const bool live = ei < n;
        const int es = live ? ei : n - 1;
        uint32_t x[G::NLL];
        {   // base -> Montgomery form, table[k] = base^k
            uint32_t bR[G::NLL], r2[G::NLL];
            load_elem<G>(bR, base + (size_t)es * base_w32, base_w32);
            load_const_slice<G>(r2, ctx->r2);
            mm_times<G>(bR, r2, lds, nm, n0inv);
#pragma unroll
            for (int j = 0; j < G::NLL; ++j) { x[j] = bR[j]; tbl(1, j) = bR[j]; }
#pragma unroll 1
            for (int k = 2; k < (1 << W); ++k) {
                mm_times<G>(x, bR, lds, nm, n0inv);
#pragma unroll
                for (int j = 0; j < G::NLL; ++j) tbl(k, j) = x[j];
            }
        }
        // top window
        {
            const uint32_t wv = exp_bits(expo, ewords, (nwin - 1) * W, W);
            if (wv == 0) load_const_slice<G>(x, ctx->one);
            else {
#pragma unroll
                for (int j = 0; j < G::NLL; ++j) x[j] = tbl((int)wv, j);
            }
        }
#pragma unroll 1
        for (int wi = nwin - 2; wi >= 0; --wi) {
            const uint32_t wv = exp_bits(expo, ewords, wi * W, W);
#pragma unroll 1
            for (int s = 0; s < W; ++s) mm_square<G>(x, lds, nm, n0inv);
            if (wv != 0) {
                uint32_t y[G::NLL];
#pragma unroll
                for (int j = 0; j < G::NLL; ++j) y[j] = tbl((int)wv, j);
                mm_times<G>(x, y, lds, nm, n0inv);
            }
        }
        if (!keep_mont) {
            uint32_t one[G::NLL];
            set_plain_one<G>(one);
            mm_times<G>(x, one, lds, nm, n0inv);
            cond_sub<G::NLL, G::T>(x, nm);
        }
        if (live) store_elem<G>(x, out + (size_t)ei * out_w32, out_w32, lds);
    }
}

// ---------------------------------------------------------------------------------------------
// Per-element exponents (e_i packed [N][EW] words, at most ebits_max significant bits): left-to-right
// binary method; the multiply step is skipped when no element of the wave needs it.
template <class G>
__global__ void __launch_bounds__(BLOCK_THREADS, 2)
k_modexp_var(const MontCtx* __restrict__ ctx, const uint32_t* __restrict__ base, int base_w32, int base_shift,
             const uint32_t* __restrict__ expo, int ew, int ebits_max, int exp_bcast,
             uint32_t* __restrict__ out, int out_w32, int n, int keep_mont, int out_raw) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    typename G::NM nm;
    load_modulus<G>(nm, ctx, lds);
    const uint32_t n0inv = ctx->n0inv;
    const int tiles = (n + G::EPB - 1) / G::EPB;
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int ei = tile * G::EPB + G::elem();
        const bool live = ei < n;
        const int es = live ? ei : n - 1;
        const uint32_t* erow = expo + (size_t)(exp_bcast ? 0 : es) * ew;
        uint32_t bR[G::NLL], x[G::NLL];
        {
            uint32_t r2[G::NLL];
            load_elem<G>(bR, base + (size_t)(es >> base_shift) * base_w32, base_w32);   // base_shift: 0 per element, 31 broadcast
            load_const_slice<G>(r2, ctx->r2);
            mm_times<G>(bR, r2, lds, nm, n0inv);
        }
        load_const_slice<G>(x, ctx->one);
        // skip the leading zero bits common to the whole wave
        int top = -1;
        for (int k = ew - 1; k >= 0 && top < 0; --k) {
            uint32_t wv = erow[k];
            int hb = wv ? (32 * k + 31 - __clz(wv)) : -1;
            if (hb >= 0) top = hb;
        }
        if (top >= ebits_max) top = ebits_max - 1;
        // wave-wide maximum of `top`
        for (int off = 32; off > 0; off >>= 1) { int o = __shfl_xor(top, off, 64); top = o > top ? o : top; }
#pragma unroll 1
        for (int bit = top; bit >= 0; --bit) {
            mm_square<G>(x, lds, nm, n0inv);
            const bool need = (erow[bit >> 5] >> (bit & 31)) & 1u;
            if (__any(need)) {
                uint32_t y[G::NLL];
#pragma unroll
                for (int j = 0; j < G::NLL; ++j) y[j] = x[j];
                mm_times<G>(y, bR, lds, nm, n0inv);
#pragma unroll
                for (int j = 0; j < G::NLL; ++j) x[j] = need ? y[j] : x[j];
            }
        }
        if (!keep_mont) {
            uint32_t one[G::NLL];
            set_plain_one<G>(one);
            mm_times<G>(x, one, lds, nm, n0inv);
            cond_sub<G::NLL, G::T>(x, nm);
        } else {
            cond_sub<G::NLL, G::T>(x, nm);               // canonical Montgomery representative
        }
        if (live) {
            if (out_raw) {                               // radix-29 limbs, NL words per element (tables)
                const int t = G::gl();
#pragma unroll
                for (int j = 0; j < G::NLL; ++j) out[(size_t)ei * G::NL + G::NLL * t + j] = x[j];
            } else {
                store_elem<G>(x, out + (size_t)ei * out_w32, out_w32, lds);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Per-element exponents with fixed windows of `wbits` bits (run-time): every lane group looks its own digit up in its
// own column of the [entry][limb][slot] scratch table (entry 0 = 1), so a multiplication costs one product per window
// instead of one per bit: 53-bit exponents 6 + 51 + 17 = 74 products at 3 bits against 53 + 53 for the binary method,
// full-size exponents (negative multipliers, n - |x|) 4 944 against 8 192.  Windows in which every element of the
// wave has digit zero are skipped.
template <class G>
__global__ void __launch_bounds__(BLOCK_THREADS, 2)
k_modexp_var_win(const MontCtx* __restrict__ ctx, const uint32_t* __restrict__ base, int base_w32,
                 const uint32_t* __restrict__ expo, int ew, int ebits_max, int exp_bcast,
                 uint32_t* __restrict__ out, int out_w32, int n, uint32_t* __restrict__ table, int wbits) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int t = G::gl();
    typename G::NM nm;
    load_modulus<G>(nm, ctx, lds);
    const uint32_t n0inv = ctx->n0inv;
    const size_t nslots = (size_t)gridDim.x * G::EPB;
    const size_t slot = (size_t)blockIdx.x * G::EPB + G::elem();
    auto tbl = [&](int entry, int j) -> uint32_t& {
        return table[((size_t)entry * G::NL + (G::NLL * t + j)) * nslots + slot];
    };
    const int NT = 1 << wbits;
    const int nwin = (ebits_max + wbits - 1) / wbits;
    const int tiles = (n + G::EPB - 1) / G::EPB;
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int ei = tile * G::EPB + G::elem();
        const bool live = ei < n;
        const int es = live ? ei : n - 1;
        const uint32_t* erow = expo + (size_t)(exp_bcast ? 0 : es) * ew;
        auto window = [&](int wi) -> int {
            const int bit = wi * wbits, k = bit >> 5;
            uint64_t bits2 = k < ew ? erow[k] : 0u;
            if (k + 1 < ew) bits2 |= (uint64_t)erow[k + 1] << 32;
            return (int)((uint32_t)(bits2 >> (bit & 31)) & (uint32_t)(NT - 1));
        };
        uint32_t x[G::NLL];
        {   // base -> Montgomery form; table[k] = base^k, table[0] = 1
            uint32_t bR[G::NLL], r2[G::NLL], one[G::NLL];
            load_elem<G>(bR, base + (size_t)es * base_w32, base_w32);
            load_const_slice<G>(r2, ctx->r2);
            load_const_slice<G>(one, ctx->one);
            mm_times<G>(bR, r2, lds, nm, n0inv);
#pragma unroll
            for (int j = 0; j < G::NLL; ++j) { x[j] = bR[j]; tbl(0, j) = one[j]; tbl(1, j) = bR[j]; }
#pragma unroll 1
            for (int k = 2; k < NT; ++k) {
                mm_times<G>(x, bR, lds, nm, n0inv);
#pragma unroll
                for (int j = 0; j < G::NLL; ++j) tbl(k, j) = x[j];
            }
        }
        {
            const int d0 = window(nwin - 1);
#pragma unroll
            for (int j = 0; j < G::NLL; ++j) x[j] = tbl(d0, j);
        }
#pragma unroll 1
        for (int wi = nwin - 2; wi >= 0; --wi) {
#pragma unroll 1
            for (int s = 0; s < wbits; ++s) mm_square<G>(x, lds, nm, n0inv);
            const int d = window(wi);
            if (__any(d != 0)) {
                uint32_t y[G::NLL];
#pragma unroll
                for (int j = 0; j < G::NLL; ++j) y[j] = tbl(d, j);
                mm_times<G>(x, y, lds, nm, n0inv);
            }
        }
        uint32_t one[G::NLL];
        set_plain_one<G>(one);
        mm_times<G>(x, one, lds, nm, n0inv);
        cond_sub<G::NLL, G::T>(x, nm);
        if (live) store_elem<G>(x, out + (size_t)ei * out_w32, out_w32, lds);
    }
}

}  // namespace pai
