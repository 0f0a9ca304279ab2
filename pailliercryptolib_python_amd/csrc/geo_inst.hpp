// Instantiates every kernel for one geometry and fills its GeoOps table.
#pragma once
#include <cstdio>
#include <cstdlib>
#include "geo_ops.hpp"
#include "kernels_modexp.hpp"

namespace pai {

template <class G>
struct GeoInst {
    // the tile-I/O kernels (k_modmul, k_pow2, k_add_aligned, k_addn) keep the modulus slice in registers
    using GM = Geo<G::NLL, G::T, G::U, false>;
    static void set_lds(const void* fn, int bytes) {
        (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    }
    // PAI_DEBUG_OCC=1: print the resident workgroups per CU the runtime computes for a kernel (dev probe)
    static void report_occupancy(const char* name, const void* fn, int bytes) {
        static const bool on = [] { const char* e = std::getenv("PAI_DEBUG_OCC"); return e && e[0] == '1'; }();
        if (!on) return;
        int nb = -1;
        hipFuncAttributes fa;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, BLOCK_THREADS, (size_t)bytes);
        (void)hipFuncGetAttributes(&fa, fn);
        fprintf(stderr, "PAI_OCC %s<%dx%d>: blocks/CU=%d lds=%d B regs=%d scratch=%zu B\n", name, G::NLL, G::T, nb, bytes, fa.numRegs,
                (size_t)fa.localSizeBytes);
    }
    // (measured and dropped, profiles/r04/ctadd_ab_*.jsonl: the same products on per-wave LDS regions with 18 limbs x 8 lanes at
    // three or four waves per SIMD: 2.14-2.35 ms per 2^20 against 1.98 ms here)
    static void modmul(hipStream_t s, int grid, const MontCtx* c, const uint32_t* a, const uint32_t* b, uint32_t* out,
                       int n, int w32, int b_bcast, int mode, const MontCtx* fin) {
        constexpr int bytes = GM::LDS_BYTES + GM::STAGE_BYTES + GM::NL * 4;     // + the R^2 copy
        if constexpr (G::T >= 16) {
            if (fin != nullptr) {        // minus-one context (c) with the true modulus' context (fin): see add_aligned
                using GA = Geo<G::NLL, G::T, G::U, false, true>;
                set_lds((const void*)k_modmul<GA>, bytes);
                hipLaunchKernelGGL(k_modmul<GA>, dim3(grid), dim3(BLOCK_THREADS), bytes, s, c, a, b, out, n, w32, b_bcast, mode, fin);
                return;
            }
        }
        set_lds((const void*)k_modmul<GM>, bytes);
        report_occupancy("k_modmul", (const void*)k_modmul<GM>, bytes);
        hipLaunchKernelGGL(k_modmul<GM>, dim3(grid), dim3(BLOCK_THREADS), bytes, s, c, a, b, out, n, w32, b_bcast, mode, nullptr);
    }
    static void modexp_fixed(hipStream_t s, int grid, const MontCtx* c, const uint32_t* base, int base_w32,
                             const uint32_t* expo, int ewords, int ebits, uint32_t* out, int out_w32, int n,
                             uint32_t* table, int keep_mont) {
        set_lds((const void*)k_modexp_fixed<G, MODEXP_WINDOW>, G::LDS_BYTES);
        hipLaunchKernelGGL((k_modexp_fixed<G, MODEXP_WINDOW>), dim3(grid), dim3(BLOCK_THREADS), G::LDS_BYTES, s, c, base,
                           base_w32, expo, ewords, ebits, out, out_w32, n, table, keep_mont);
    }
    static void modexp_var(hipStream_t s, int grid, const MontCtx* c, const uint32_t* base, int base_w32, int base_shift,
                           const uint32_t* expo, int ew, int ebits_max, int exp_bcast, uint32_t* out, int out_w32,
                           int n, int keep_mont, int out_raw) {
        set_lds((const void*)k_modexp_var<G>, G::LDS_BYTES);
        hipLaunchKernelGGL(k_modexp_var<G>, dim3(grid), dim3(BLOCK_THREADS), G::LDS_BYTES, s, c, base, base_w32,
                           base_shift, expo, ew, ebits_max, exp_bcast, out, out_w32, n, keep_mont, out_raw);
    }
    static void encrypt(hipStream_t s, int grid, EncParams P, const uint32_t* m, const uint32_t* r,
                        const uint32_t* ct_in, uint32_t* ct_out, int n, int mode) {
        if constexpr (G::T >= 16 && G::T <= 64) {
            using GM1 = Geo<G::NLL, G::T, G::U, false, true>;
            if (mode == 7) {       // table conversion to a minus-one context's Montgomery form: m = table in, ct_out = table out, r = the constant
                set_lds((const void*)k_fb_to_m1<GM1>, GM1::LDS_BYTES);
                hipLaunchKernelGGL(k_fb_to_m1<GM1>, dim3(grid), dim3(BLOCK_THREADS), GM1::LDS_BYTES, s, P.nsq, m, ct_out, n, r);
                return;
            }
            if ((mode == 3 || mode == 0) && P.fin != nullptr) {     // ct + plaintext / raw encryption of small batches on a minus-one context
                set_lds((const void*)k_encrypt<GM1>, GM1::LDS_BYTES);
                hipLaunchKernelGGL(k_encrypt<GM1>, dim3(grid), dim3(BLOCK_THREADS), GM1::LDS_BYTES, s, P, m, r, ct_in, ct_out, n, mode);
                return;
            }
            if (mode >= 5 && P.fin != nullptr) {     // the shared chain on a minus-one context
                set_lds((const void*)k_encrypt_tree<GM1>, GM1::LDS_BYTES);
                hipLaunchKernelGGL(k_encrypt_tree<GM1>, dim3(grid), dim3(BLOCK_THREADS), GM1::LDS_BYTES, s, P, m, r, ct_in, ct_out, n, mode - 4);
                return;
            }
            if (mode >= 5) {       // modes 5 / 6: modes 1 / 2 with the fixed-base chain shared by the four waves (grid counts 64 / T integers)
                set_lds((const void*)k_encrypt_tree<G>, G::LDS_BYTES);
                hipLaunchKernelGGL(k_encrypt_tree<G>, dim3(grid), dim3(BLOCK_THREADS), G::LDS_BYTES, s, P, m, r, ct_in, ct_out, n, mode - 4);
                return;
            }
        }
        set_lds((const void*)k_encrypt<G>, G::LDS_BYTES);
        hipLaunchKernelGGL(k_encrypt<G>, dim3(grid), dim3(BLOCK_THREADS), G::LDS_BYTES, s, P, m, r, ct_in, ct_out, n, mode);
    }
    static void pair_finish(hipStream_t s, int grid, EncParams P, const uint32_t* wv, int wv_words, const uint32_t* ct_in,
                            uint32_t* ct_out, int n, int mul_ct) {
        set_lds((const void*)k_pair_finish<G>, G::LDS_BYTES);
        hipLaunchKernelGGL(k_pair_finish<G>, dim3(grid), dim3(BLOCK_THREADS), G::LDS_BYTES, s, P, wv, wv_words, ct_in, ct_out, n, mul_ct);
    }
    static void fb_expand(hipStream_t s, int grid, const MontCtx* c, const uint32_t* S, uint32_t* T, int J, int h) {
        set_lds((const void*)k_fb_expand<G>, G::LDS_BYTES);
        hipLaunchKernelGGL(k_fb_expand<G>, dim3(grid), dim3(BLOCK_THREADS), G::LDS_BYTES, s, c, S, T, J, h);
    }
    static void dec_a(hipStream_t s, int gridx, DecAParams P, const uint32_t* ct, uint32_t* u_out, int n,
                      uint32_t* table) {
        // the wide-group (latency) geometries run stage A on minus-one contexts (mont_dev.hpp: Rows::block_m1)
        using GD = Geo<G::NLL, G::T, G::U, G::NMLDS, (G::T >= 16)>;
        if constexpr (GD::M1) {
            if (P.rl) {         // gridx counts workgroups of EPB / 2 integers here
                constexpr int bytes = ((RL_RING + 1) * GD::LDS_WORDS + 16) * 4;
                set_lds((const void*)k_dec_a_rl<GD>, bytes);
                hipLaunchKernelGGL((k_dec_a_rl<GD>), dim3(gridx, 2), dim3(BLOCK_THREADS), bytes, s, P, ct, u_out, n);
                return;
            }
        }
        set_lds((const void*)k_dec_a<GD, MODEXP_WINDOW>, GD::LDS_BYTES);
        hipLaunchKernelGGL((k_dec_a<GD, MODEXP_WINDOW>), dim3(gridx, 2), dim3(BLOCK_THREADS), GD::LDS_BYTES, s, P, ct,
                           u_out, n, table);
    }
    static void dec_b(hipStream_t s, int grid, DecBParams P, const uint32_t* u_in, uint32_t* m_out, int n) {
        constexpr int bytes = 2 * G::LDS_WORDS * 4;
        set_lds((const void*)k_dec_b<G>, bytes);
        hipLaunchKernelGGL(k_dec_b<G>, dim3(grid), dim3(BLOCK_THREADS), bytes, s, P, u_in, m_out, n);
    }
    static void modexp_var_win(hipStream_t s, int grid, const MontCtx* c, const uint32_t* base, int base_w32, const uint32_t* expo,
                               int ew, int ebits_max, int exp_bcast, uint32_t* out, int out_w32, int n, uint32_t* table, int wbits,
                               const MontCtx* fin) {
        // 8-lane geometries: modulus slice from LDS (3072 / 4096-bit ct * pt 20.8 -> 20.5 / 32.7 -> 31.8 ms per 65536)
        using GV = Geo<G::NLL, G::T, G::U, (G::T >= 8 && G::NLL % 4 == 0)>;
        if constexpr (G::T >= 16) {
            if (fin != nullptr) {        // wide-group geometries with a minus-one context (c) and the true modulus' context (fin)
                using GM1 = Geo<G::NLL, G::T, G::U, false, true>;
                if (wbits == 0) {    // right to left on wave pairs, no table (grid counts workgroups of EPB / 2 integers)
                    constexpr int bytes = ((RL_RING + 1) * GM1::LDS_WORDS + 16) * 4;
                    set_lds((const void*)k_modexp_rl<GM1>, bytes);
                    hipLaunchKernelGGL(k_modexp_rl<GM1>, dim3(grid), dim3(BLOCK_THREADS), bytes, s, c, base, base_w32, expo, ew, ebits_max,
                                       exp_bcast, out, out_w32, n, fin);
                    return;
                }
                set_lds((const void*)k_modexp_var_win<GM1>, VarWinCfg<GM1>::LDS_BYTES);
                hipLaunchKernelGGL(k_modexp_var_win<GM1>, dim3(grid), dim3(BLOCK_THREADS), VarWinCfg<GM1>::LDS_BYTES, s, c, base, base_w32,
                                   expo, ew, ebits_max, exp_bcast, out, out_w32, n, table, wbits, fin);
                return;
            }
        }
        set_lds((const void*)k_modexp_var_win<GV>, VarWinCfg<GV>::LDS_BYTES);
        hipLaunchKernelGGL(k_modexp_var_win<GV>, dim3(grid), dim3(BLOCK_THREADS), VarWinCfg<GV>::LDS_BYTES, s, c, base, base_w32, expo, ew,
                           ebits_max, exp_bcast, out, out_w32, n, table, wbits, (const MontCtx*)nullptr);
    }
    static void sq_chain(hipStream_t s, const MontCtx* c, const MontCtx* fin, const uint32_t* base, int w32, uint32_t* out, int h,
                         int nsnap) {
        if constexpr (G::T >= 16) {
            if (fin != nullptr) {
                using GM1 = Geo<G::NLL, G::T, G::U, false, true>;
                set_lds((const void*)k_sq_chain<GM1>, GM1::LDS_BYTES);
                hipLaunchKernelGGL(k_sq_chain<GM1>, dim3(1), dim3(BLOCK_THREADS), GM1::LDS_BYTES, s, c, fin, base, w32, out, h, nsnap);
                return;
            }
        }
        set_lds((const void*)k_sq_chain<G>, G::LDS_BYTES);
        hipLaunchKernelGGL(k_sq_chain<G>, dim3(1), dim3(BLOCK_THREADS), G::LDS_BYTES, s, c, (const MontCtx*)nullptr, base, w32, out, h, nsnap);
    }
    static void pow2(hipStream_t s, int grid, const MontCtx* c, uint32_t* ct, const int32_t* delta, int delta_bcast,
                     int n, int w32, const MontCtx* fin) {
        constexpr int bytes = GM::LDS_BYTES + GM::STAGE_BYTES + GM::NL * 4;     // + the R^2 copy
        if constexpr (G::T >= 16) {
            if (fin != nullptr) {        // minus-one context (c) with the true modulus' context (fin): see add_aligned
                using GA = Geo<G::NLL, G::T, G::U, false, true>;
                set_lds((const void*)k_pow2<GA>, bytes);
                hipLaunchKernelGGL(k_pow2<GA>, dim3(grid), dim3(BLOCK_THREADS), bytes, s, c, ct, delta, delta_bcast, n, w32, fin);
                return;
            }
        }
        set_lds((const void*)k_pow2<GM>, bytes);
        hipLaunchKernelGGL(k_pow2<GM>, dim3(grid), dim3(BLOCK_THREADS), bytes, s, c, ct, delta, delta_bcast, n, w32, nullptr);
    }
    static void add_aligned(hipStream_t s, int grid, const MontCtx* c, const uint32_t* a, const uint32_t* b, int b_bcast,
                            const int32_t* delta, uint32_t* out, int n, int w32, const uint32_t* entry, const MontCtx* fin) {
        constexpr int bytes = GM::LDS_BYTES + GM::STAGE_BYTES + GM::NL * 4;
        if constexpr (G::T >= 16) {
            if (fin != nullptr) {        // wide-group geometries with a minus-one context (c) and the true modulus' context (fin)
                using GA = Geo<G::NLL, G::T, G::U, false, true>;
                set_lds((const void*)k_add_aligned<GA>, bytes);
                hipLaunchKernelGGL(k_add_aligned<GA>, dim3(grid), dim3(BLOCK_THREADS), bytes, s, c, a, b, b_bcast, delta, out, n, w32, entry, fin);
                return;
            }
        }
        set_lds((const void*)k_add_aligned<GM>, bytes);
        hipLaunchKernelGGL(k_add_aligned<GM>, dim3(grid), dim3(BLOCK_THREADS), bytes, s, c, a, b, b_bcast, delta, out, n, w32, entry, nullptr);
    }
    static void addn(hipStream_t s, int grid, const MontCtx* c, AddnArgs A, uint32_t* out, int n, int w32, const uint32_t* rpow) {
        constexpr int bytes = GM::LDS_BYTES + GM::STAGE_BYTES + GM::NL * 4;     // + the domain-entry constant
        set_lds((const void*)k_addn<GM>, bytes);
        report_occupancy("k_addn", (const void*)k_addn<GM>, bytes);
        hipLaunchKernelGGL(k_addn<GM>, dim3(grid), dim3(BLOCK_THREADS), bytes, s, c, A, out, n, w32, rpow);
    }
    static void mexp_table(hipStream_t s, int grid, const MontCtx* c, const uint32_t* ct, const uint32_t* ct_inv, int w32,
                           uint32_t* table, int nentries, int nsigns, int wbits) {
        set_lds((const void*)k_mexp_table<G>, G::LDS_BYTES);
        hipLaunchKernelGGL(k_mexp_table<G>, dim3(grid), dim3(BLOCK_THREADS), G::LDS_BYTES, s, c, ct, ct_inv, w32, table, nentries, nsigns, wbits);
    }
    static void mexp(hipStream_t s, int grid, const MontCtx* c, MexpParams P, const uint32_t* table, const uint32_t* e,
                     const uint8_t* sign, uint32_t* out, int nlanes) {
        using GX = Geo<G::NLL, G::T, G::U, false>;
        set_lds((const void*)k_mexp<GX>, GX::LDS_BYTES);
        hipLaunchKernelGGL(k_mexp<GX>, dim3(grid), dim3(BLOCK_THREADS), GX::LDS_BYTES, s, c, P, table, e, sign, out, nlanes);
    }
    static void modmul_msb(hipStream_t s, int grid, const MsbCtx* c, const uint32_t* a, const uint32_t* b, uint32_t* out, int n, int w32) {
        if constexpr (G::T <= 8) {
            // (rows per block: the geometry's own — 9 / 12 / 18 on 36 x 4 spill inside the row loop, /tmp probe of round 6)
            constexpr int bytes = MsbLds<GM>::WORDS * 4;
            set_lds((const void*)k_modmul_msb<GM>, bytes);
            report_occupancy("k_modmul_msb", (const void*)k_modmul_msb<GM>, bytes);
            hipLaunchKernelGGL(k_modmul_msb<GM>, dim3(grid), dim3(BLOCK_THREADS), bytes, s, c, a, b, out, n, w32);
        }
    }
    static size_t table_words(size_t blocks) { return (size_t)(1u << MODEXP_WINDOW) * G::NL * blocks * G::EPB; }

    static const GeoOps* ops() {
        static const GeoOps o = {G::NLL, G::T, G::U, G::NL, G::EPB, G::LDS_BYTES, 2 * G::LDS_WORDS * 4,
                                 &modmul, &modexp_fixed, &modexp_var, &modexp_var_win, &encrypt, &fb_expand, &dec_a, &dec_b, &pow2, &sq_chain, &add_aligned, &addn, &table_words, &pair_finish, &mexp_table, &mexp,
                                 G::T <= 8 ? &modmul_msb : nullptr};
        return &o;
    }
};

}  // namespace pai
