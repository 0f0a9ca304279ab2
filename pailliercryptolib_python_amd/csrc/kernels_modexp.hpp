// Generic batched modular kernels on an arbitrary odd modulus (MontCtx):
//   k_modmul        out_i = a_i * b_i mod M                  (ct+ct add: CipherText::operator+, classes.cpp:318-321)
//   k_modexp_fixed  out_i = base_i ^ E mod M, E wave-uniform  (CRT-decrypt halves, standard obfuscator r^n)
//   k_modexp_var    out_i = base_i ^ e_i mod M                (ct*pt: CipherText::operator*, classes.cpp:324-325)
// All operands are packed little-endian u32 words, row-major [N][W32]; results are canonical residues.
#pragma once
#include "kernels_common.hpp"
#include "mont_msb.hpp"

namespace pai {

// ---------------------------------------------------------------------------------------------
// out_i = a_i * b_i mod M on canonical packed rows.  Every wave works on its own tiles of 64 / T consecutive
// elements: coalesced full-width copies HBM <-> LDS (load_tile / store_tile), one or two Montgomery products, no
// workgroup barrier anywhere in the loop, so the waves drift apart and the two waves of a SIMD overlap one's
// memory phase with the other's multiply phase.
//   mode MODMUL_FULL   out = a*b mod M          (a*b*R^-1, then * R^2 * R^-1; b_bcast: b*R is formed once per wave,
//                                               so a broadcast addend costs ONE product per element)
//   mode MODMUL_MONT   out = a*b*R^-1 mod M     (canonical residue of the Montgomery product: the body of the
//                                               product trees of pai_ct_invert / pai_ct_prod, which keep track of
//                                               the power of R per tree level on the host)
template <class G>
__global__ void __launch_bounds__(BLOCK_THREADS, 2)
k_modmul(const MontCtx* __restrict__ ctx, const uint32_t* a, const uint32_t* b, uint32_t* out,
         int n, int w32, int b_bcast, int mode, const MontCtx* __restrict__ fin) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    using WT = WaveTile<G>;
    uint32_t* stage = lds + G::LDS_WORDS + G::NL;        // behind the operand buffer and the modulus copy
    uint32_t* r2_lds = stage + G::STAGE_WORDS;           // R^2 mod M, one copy per workgroup
    typename G::NM nm;
    load_modulus<G>(nm, ctx, lds);
    for (int i = threadIdx.x; i < G::NL; i += BLOCK_THREADS) r2_lds[i] = ctx->r2[i];
    __syncthreads();
    // minus-one contexts (G::M1: small wire-form batches on one integer per wavefront, MODMUL_FULL without broadcast only — as
    // k_add_aligned): ctx is the context of M k, its scalar the number of row blocks, fin the context of M itself for the way out
    const uint32_t n0inv = G::M1 ? ctx->rows / G::U : ctx->n0inv;
    constexpr int WPB = BLOCK_THREADS / 64;
    const int wtiles = (n + WT::EPW - 1) / WT::EPW;
    clear_stage<G>(stage);
    // Register plan: the left operand x lives in registers (the row engine's `a`), the right operand streams from
    // LDS — the per-element operand buffer for b, the workgroup's R^2 copy for the domain fix-up — so nothing but x,
    // the modulus slice and the accumulator window is live inside the row loops (no scratch traffic there).
    const uint32_t* b_lds = lds + G::elem();             // column of this element in the [limb][element] buffer
    if (b_bcast) {                                       // the shared operand, staged once per wave for all its elements
        uint32_t bb[G::NLL];
        load_tile<G>(stage, b, 1, w32, true);
        unpack_row<G>(bb, stage);
        if (mode == MODMUL_FULL) {                       // b*R (lazy, < 2M): a broadcast addend then costs ONE product per element
            uint32_t t[G::NLL];
            if constexpr (G::M1) mont_mul_m1<G::NLL, G::U, G::T>(t, bb, r2_lds, 1, nm, (int)n0inv);
            else mont_mul<G::NLL, G::U, G::T>(t, bb, r2_lds, 1, nm, n0inv);
#pragma unroll
            for (int j = 0; j < G::NLL; ++j) bb[j] = t[j];
        }
        stage_b<G>(bb, lds);
    }
    // the products of one tile run through ONE rolled loop body (pass 0: the operand pair; pass 1 of MODMUL_FULL
    // without broadcast: the domain fix-up by R^2): a second inlined copy of the row engine was measured at 3-4x
    // the cycles of the first (register allocation across two copies spills inside the row loops)
    const int npass = (mode == MODMUL_FULL && !b_bcast) ? 2 : 1;
    // every wave streams through ONE contiguous run of tiles
    const int per_wave = (wtiles + (int)gridDim.x * WPB - 1) / ((int)gridDim.x * WPB);
    const int wt_begin = ((int)blockIdx.x * WPB + WT::wave()) * per_wave;
    const int wt_end = min(wtiles, wt_begin + per_wave);
    for (int wt = wt_begin; wt < wt_end; ++wt) {
        const int row0 = wt * WT::EPW;
        const int rows = min(WT::EPW, n - row0);
        uint32_t x[G::NLL];
        // The tile I/O phases are short on VALU work and long on memory latency: they run at raised priority so that
        // the multiply stream of the other wave on this SIMD does not starve their address arithmetic (VALU
        // arbitration is priority, then age); the products run at base priority.
        __builtin_amdgcn_s_setprio(2);
        {
            if (!b_bcast) {
                load_tile<G>(stage, b + (size_t)row0 * w32, rows, w32);
                unpack_row<G>(x, stage);
                stage_b<G>(x, lds);
            }
            load_tile<G>(stage, a + (size_t)row0 * w32, rows, w32);
        }
        unpack_row<G>(x, stage);
        __builtin_amdgcn_s_setprio(0);
#pragma unroll 1
        for (int pass = 0; pass < npass; ++pass) {
            uint32_t r[G::NLL];
            // pass 0: a*b*R^-1 (a*b when the staged operand is b*R); pass 1: * R^2 * R^-1
            if constexpr (G::M1) mont_mul_m1<G::NLL, G::U, G::T>(r, x, pass == 0 ? b_lds : r2_lds, pass == 0 ? G::EPB : 1, nm, (int)n0inv);
            else mont_mul<G::NLL, G::U, G::T>(r, x, pass == 0 ? b_lds : r2_lds, pass == 0 ? G::EPB : 1, nm, n0inv);
#pragma unroll
            for (int j = 0; j < G::NLL; ++j) x[j] = r[j];
        }
        if constexpr (G::M1) m1_reduce_to_true_modulus<G>(x, lds, fin);
        else cond_sub<G::NLL, G::T>(x, nm);
        __builtin_amdgcn_s_setprio(2);
        pack_row<G>(x, stage);
        store_tile<G>(stage, out + (size_t)row0 * w32, rows, w32);
    }
    __builtin_amdgcn_s_setprio(0);
}

// ---------------------------------------------------------------------------------------------
// out_i = a_i * b_i mod M on canonical packed rows by ONE most-significant-limb-first product (mont_msb.hpp) — k_modmul's
// MODMUL_FULL without a broadcast operand at half the limb products.  Same tile loop, same LDS plan; the modulus copy
// behind the operand buffer holds Mt = M B^off for the two conditional subtractions, and the result's B^off leaves
// through the operand buffer (limb j + off read back as limb j).  Operands may be any word pattern of the row.
template <class G>
struct MsbLds {                                          // LDS plan of k_modmul_msb, in words
    static constexpr int OP = (G::NL + MSB_OFF_MAX) * G::EPB;      // operand buffer [limb][element] + MSB_OFF_MAX limbs of zeros behind it
    static constexpr int MT = OP, STAGE = MT + G::NL, W = STAGE + G::STAGE_WORDS, WORDS = W + G::NL;
};

template <class G, int UM = G::U>
__global__ void __launch_bounds__(BLOCK_THREADS, 2)
k_modmul_msb(const MsbCtx* __restrict__ ctx, const uint32_t* a, const uint32_t* b, uint32_t* out, int n, int w32) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    using WT = WaveTile<G>;
    using L = MsbLds<G>;
    uint32_t* mt_lds = lds + L::MT;                      // Mt = M B^off, for the conditional subtraction
    uint32_t* stage = lds + L::STAGE;
    uint32_t* w_lds = lds + L::W;                        // W = B^NL - Mt, one copy per workgroup
    const int t = G::gl();
    // W's slice: registers where the accumulator window leaves room, else re-read from LDS during the q W step (the window, a's
    // slice and W's are 156 of the 256 registers at 36 limbs per lane: with W in registers the compiler spills loop invariants)
    constexpr bool W_LDS = G::NLL % 4 == 0 && G::NLL >= 32;
    typename std::conditional<W_LDS, NmLds<G::NLL>, NmRegs<G::NLL>>::type wm;
    if constexpr (W_LDS) wm.p = w_lds + G::NLL * t;
    else {
#pragma unroll
        for (int j = 0; j < G::NLL; ++j) wm.v[j] = ctx->w[G::NLL * t + j];
    }
    for (int i = threadIdx.x; i < G::NL; i += BLOCK_THREADS) { mt_lds[i] = ctx->mt[i]; w_lds[i] = ctx->w[i]; }
    for (int i = threadIdx.x; i < L::OP; i += BLOCK_THREADS) lds[i] = 0u;
    __syncthreads();
    NmLds<G::NLL> mt;
    mt.p = mt_lds + G::NLL * t;
    const uint32_t tb = ctx->tb;
    // The multiplier enters as b B^off and the result leaves as r B^off: limb j is staged as limb j + off and read back from
    // there — ONE base address and compile-time offsets, no bounds to test: what lands in the MSB_OFF_MAX limbs behind the
    // buffer is zero (b fits the row's words), and the lowest `off` limbs stay zero from tile to tile (the buffer was cleared; a
    // tile leaves r B^off, a multiple of B^off, behind).
    uint32_t* opsh = lds + (G::NLL * t + (int)ctx->off) * G::EPB + G::elem();
    MsbK k;
    k.mu = ctx->mu;
    k.sh1 = tb - 3u;
    k.sh2 = 32u - tb;
    k.sh3 = (uint32_t)RB - tb;
    k.m2 = ctx->m2;
    k.eight = ctx->eight;
    constexpr int WPB = BLOCK_THREADS / 64;
    const int wtiles = (n + WT::EPW - 1) / WT::EPW;
    clear_stage<G>(stage);
    const uint32_t* b_lds = lds + G::elem();
    const int per_wave = (wtiles + (int)gridDim.x * WPB - 1) / ((int)gridDim.x * WPB);
    const int wt_begin = ((int)blockIdx.x * WPB + WT::wave()) * per_wave;
    const int wt_end = min(wtiles, wt_begin + per_wave);
    for (int wt = wt_begin; wt < wt_end; ++wt) {
        const int row0 = wt * WT::EPW;
        const int rows = min(WT::EPW, n - row0);
        uint32_t x[G::NLL];
        __builtin_amdgcn_s_setprio(2);
        load_tile<G>(stage, b + (size_t)row0 * w32, rows, w32);
        unpack_row<G>(x, stage);
        wave_lds_fence();
#pragma unroll
        for (int j = 0; j < G::NLL; ++j) opsh[j * G::EPB] = x[j];
        wave_lds_fence();
        load_tile<G>(stage, a + (size_t)row0 * w32, rows, w32);
        unpack_row<G>(x, stage);
        __builtin_amdgcn_s_setprio(0);
        {
            uint32_t r[G::NLL];
            msb_mul<G::NLL, UM, G::T>(r, x, b_lds, G::EPB, wm, k);
#pragma unroll
            for (int j = 0; j < G::NLL; ++j) x[j] = r[j];
        }
        cond_sub<G::NLL, G::T>(x, mt);
        __builtin_amdgcn_s_setprio(2);
        stage_b<G>(x, lds);                                  // / B^off
#pragma unroll
        for (int j = 0; j < G::NLL; ++j) x[j] = opsh[j * G::EPB];
        pack_row<G>(x, stage);
        store_tile<G>(stage, out + (size_t)row0 * w32, rows, w32);
    }
    __builtin_amdgcn_s_setprio(0);
}

// ---------------------------------------------------------------------------------------------
// Fixed-window (W bits) exponentiation with a wave-uniform exponent.  The 2^W-entry table of each
// resident element lives in a global scratch area laid out [entry][limb][slot] so that a wave reads
// one entry with coalesced 128/256-byte rows; slot = blockIdx.x*EPB + element.
template <class G, int W>
__global__ void __launch_bounds__(BLOCK_THREADS, PAI_LG_WAVES(G))
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
__global__ void __launch_bounds__(BLOCK_THREADS, PAI_LG_WAVES(G))
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
//
// Lane groups of 4 and 8 (keys above 2048 bits: BASELINE configs 4 and 5) keep the table ROW-major per slot —
// [slot][entry][NL], every lane reading / writing its contiguous slice with 16-byte accesses — and never hold a table
// entry in registers: the entry of the coming multiplication is STREAMED into a second LDS operand buffer, four words per
// row block, while the window's last squaring runs (RowStream, mont_dev.hpp), and the multiplication takes it from
// there as its right operand.  (Round 2 gathered 36 single dwords per lane from the [entry][limb][slot] layout right
// before each multiplication: 27 GB fetched for 67 MB of ciphertexts, 27 % of the wave cycles waiting —
// profiles/r02/pmc_k4096_r02.json.)
template <class G>
struct VarWinCfg {
    static constexpr bool STREAM = (G::T == 4 || G::T == 8) && G::NLL % 4 == 0;
    static constexpr int BUF1 = G::LDS_WORDS + G::NL;                     // second operand buffer, behind the modulus copy
    static constexpr int LDS_BYTES = G::LDS_BYTES + (STREAM ? G::LDS_WORDS * 4 : 0);
};

template <class G>
__global__ void __launch_bounds__(BLOCK_THREADS, PAI_VARWIN_WAVES(G))
k_modexp_var_win(const MontCtx* __restrict__ ctx, const uint32_t* __restrict__ base, int base_w32,
                 const uint32_t* __restrict__ expo, int ew, int ebits_max, int exp_bcast,
                 uint32_t* __restrict__ out, int out_w32, int n, uint32_t* __restrict__ table, int wbits,
                 const MontCtx* __restrict__ fin) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int t = G::gl();
    typename G::NM nm;
    load_modulus<G>(nm, ctx, lds);
    // minus-one geometries (small batches on one integer per wavefront): ctx is the context of M k, the scalar next to the
    // modulus slice is the number of row blocks, and `fin` (M's own context) reduces the result at the end
    const uint32_t n0inv = G::M1 ? ctx->rows / G::U : ctx->n0inv;
    if constexpr (VarWinCfg<G>::STREAM) {
        constexpr int CH = 4, NCH = G::NLL / 4;
        using RS = RowStream<CH, NCH, 0>;
        const int NT = 1 << wbits;
        const int nwin = (ebits_max + wbits - 1) / wbits;
        const size_t slot = (size_t)blockIdx.x * G::EPB + G::elem();
        uint32_t* trow = table + slot * (size_t)NT * G::NL + G::NLL * t;       // this lane's slice of entry 0
        auto tload = [&](uint32_t (&v)[G::NLL], int entry) {
            const uint4* p4 = reinterpret_cast<const uint4*>(trow + (size_t)entry * G::NL);
#pragma unroll
            for (int c = 0; c < NCH; ++c) { const uint4 q = p4[c]; v[4 * c] = q.x; v[4 * c + 1] = q.y; v[4 * c + 2] = q.z; v[4 * c + 3] = q.w; }
        };
        auto tstore = [&](const uint32_t (&v)[G::NLL], int entry) {
            uint4* p4 = reinterpret_cast<uint4*>(trow + (size_t)entry * G::NL);
#pragma unroll
            for (int c = 0; c < NCH; ++c) p4[c] = make_uint4(v[4 * c], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3]);
        };
        const int col = (G::NLL * t) * G::EPB + G::elem();
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
                uint32_t bR[G::NLL], c[G::NLL];
                load_elem<G>(bR, base + (size_t)es * base_w32, base_w32);
                load_const_slice<G>(c, ctx->r2);
                mm_times<G>(bR, c, lds, nm, n0inv);
                load_const_slice<G>(c, ctx->one);
                tstore(c, 0);
                tstore(bR, 1);
#pragma unroll
                for (int j = 0; j < G::NLL; ++j) x[j] = bR[j];
#pragma unroll 1
                for (int k = 2; k < NT; ++k) {
                    mm_times<G>(x, bR, lds, nm, n0inv);
                    tstore(x, k);
                }
            }
            tload(x, window(nwin - 1));
#pragma unroll 1
            for (int wi = nwin - 2; wi >= 0; --wi) {
                const int d = window(wi);
                const bool any = __any(d != 0);
                RS pf;
                pf.src0 = trow + (size_t)d * G::NL;
                pf.src1 = pf.src0;
                pf.dst0 = lds + VarWinCfg<G>::BUF1 + col;
                pf.dst1 = pf.dst0;
                pf.stride = G::EPB;
                // one rolled body for the window's products: wbits squarings (x staged as its own right operand), then —
                // if some element of the wave has a non-zero digit — the multiplication by the streamed table entry
                const int nsteps = wbits + (any ? 1 : 0);
#pragma unroll 1
                for (int s = 0; s < nsteps; ++s) {
                    const bool is_mul = s == wbits;
                    if (!is_mul) stage_b<G>(x, lds);
                    else wave_lds_fence();
                    pf.on = any && s == wbits - 1;
                    uint32_t r[G::NLL];
                    mont_mul<G::NLL, G::U, G::T>(r, x, lds + (is_mul ? VarWinCfg<G>::BUF1 : 0) + G::elem(), G::EPB, nm, n0inv, &pf);
#pragma unroll
                    for (int j = 0; j < G::NLL; ++j) x[j] = r[j];
                }
            }
            uint32_t one[G::NLL];
            set_plain_one<G>(one);
            mm_times<G>(x, one, lds, nm, n0inv);
            cond_sub<G::NLL, G::T>(x, nm);
            if (live) store_elem<G>(x, out + (size_t)ei * out_w32, out_w32, lds);
        }
        return;
    }
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
        if constexpr (G::M1) m1_reduce_to_true_modulus<G>(x, lds, fin);
        else cond_sub<G::NLL, G::T>(x, nm);
        if (live) store_elem<G>(x, out + (size_t)ei * out_w32, out_w32, lds);
    }
}

// ---------------------------------------------------------------------------------------------
// Multi-exponentiation on the lane-group engine (keys the base-n digit engine does not serve; see k_mexp_padic in
// kernels_padic_enc.hpp for the scheme): per (base, sign) a table of the powers 0 .. 2^wbits - 1 in Montgomery form (raw
// radix-29 rows of NL limbs), then one lane group per (chunk of members, output element) with one chain of squarings.
template <class G>
__global__ void __launch_bounds__(BLOCK_THREADS, PAI_LG_WAVES(G))
k_mexp_table(const MontCtx* __restrict__ ctx, const uint32_t* __restrict__ ct, const uint32_t* __restrict__ ct_inv, int w32,
             uint32_t* __restrict__ table, int nentries, int nsigns, int wbits) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int t = G::gl();
    typename G::NM nm;
    load_modulus<G>(nm, ctx, lds);
    const uint32_t n0inv = ctx->n0inv;
    const int NT = 1 << wbits;
    const int tiles = (nentries + G::EPB - 1) / G::EPB;
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int idx = tile * G::EPB + G::elem();
        const bool live = idx < nentries;
        const int is = live ? idx : nentries - 1;
        const int b = is / nsigns, sg = is - b * nsigns;
        uint32_t* ent = table + (size_t)is * NT * G::NL + G::NLL * t;
        uint32_t bR[G::NLL], x[G::NLL], c[G::NLL];
        load_elem<G>(bR, (sg ? ct_inv : ct) + (size_t)b * w32, w32);
        load_const_slice<G>(c, ctx->r2);
        mm_times<G>(bR, c, lds, nm, n0inv);
        load_const_slice<G>(c, ctx->one);
#pragma unroll
        for (int j = 0; j < G::NLL; ++j) x[j] = bR[j];
        if (live) {
#pragma unroll
            for (int j = 0; j < G::NLL; ++j) { ent[j] = c[j]; ent[G::NL + j] = bR[j]; }
        }
#pragma unroll 1
        for (int k = 2; k < NT; ++k) {
            mm_times<G>(x, bR, lds, nm, n0inv);
            if (live) {
#pragma unroll
                for (int j = 0; j < G::NLL; ++j) ent[(size_t)k * G::NL + j] = x[j];
            }
        }
    }
}

template <class G>
__global__ void __launch_bounds__(BLOCK_THREADS, PAI_MEXP_WAVES(G))
k_mexp(const MontCtx* __restrict__ ctx, MexpParams P, const uint32_t* __restrict__ table, const uint32_t* __restrict__ e,
       const uint8_t* __restrict__ sign, uint32_t* __restrict__ out, int nlanes) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int t = G::gl();
    typename G::NM nm;
    load_modulus<G>(nm, ctx, lds);
    const uint32_t n0inv = ctx->n0inv;
    const int W = P.wbits, NT = 1 << W;
    const int nwin = (P.ebits_max + W - 1) / W;
    const int Gn = P.R * P.M;
    const int tiles = (nlanes + G::EPB - 1) / G::EPB;
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int idx = tile * G::EPB + G::elem();
        const bool live = idx < nlanes;
        const int is = live ? idx : nlanes - 1;
        const int ch = is / Gn, g = is - ch * Gn, r = g / P.M, j = g - r * P.M;
        const int l0 = ch * P.chunk, l1 = min(P.K, l0 + P.chunk);
        uint32_t x[G::NLL];
        load_const_slice<G>(x, ctx->one);
        bool started = false;                         // wave-uniform: only ones so far
#pragma unroll 1
        for (int wi = nwin - 1; wi >= 0; --wi) {
            if (started) {
#pragma unroll 1
                for (int sq = 0; sq < W; ++sq) mm_square<G>(x, lds, nm, n0inv);
            }
            const int bit = wi * W, k = bit >> 5, sh = bit & 31;
#pragma unroll 1
            for (int li = 0; li < P.chunk; ++li) {
                const int l = l0 + li;
                const bool has = live && l < l1;
                const int ls = l < P.K ? l : P.K - 1;
                const size_t eoff = (((size_t)r * P.K + ls) * P.M + j) * P.e_words;
                uint64_t bits2 = k < P.e_words ? e[eoff + k] : 0u;
                if (k + 1 < P.e_words) bits2 |= (uint64_t)e[eoff + k + 1] << 32;
                const int d = has ? (int)((uint32_t)(bits2 >> sh) & (uint32_t)(NT - 1)) : 0;
                if (__any(d != 0)) {
                    const int sg = (sign && P.nsigns > 1) ? (int)sign[(size_t)ls * P.M + j] : 0;
                    const uint32_t* ent = table + ((((size_t)r * P.K + ls) * P.nsigns + sg) * NT + d) * G::NL + G::NLL * t;
                    uint32_t y[G::NLL];
                    if constexpr (G::NLL % 4 == 0) {                 // limb slices move as 16-byte vectors
                        const uint4* e4 = reinterpret_cast<const uint4*>(ent);
#pragma unroll
                        for (int c = 0; c < G::NLL / 4; ++c) {
                            const uint4 v = e4[c];
                            y[4 * c] = v.x; y[4 * c + 1] = v.y; y[4 * c + 2] = v.z; y[4 * c + 3] = v.w;
                        }
                    } else {
#pragma unroll
                        for (int jj = 0; jj < G::NLL; ++jj) y[jj] = ent[jj];
                    }
                    mm_times<G>(x, y, lds, nm, n0inv);
                    started = true;
                }
            }
        }
        uint32_t one[G::NLL];
        set_plain_one<G>(one);
        mm_times<G>(x, one, lds, nm, n0inv);
        cond_sub<G::NLL, G::T>(x, nm);
        if (live) store_elem<G>(x, out + (size_t)idx * P.w32, P.w32, lds);
    }
}

}  // namespace pai
