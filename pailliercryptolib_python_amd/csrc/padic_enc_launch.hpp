// Launchers of the base-n digit-pair kernels (kernels_padic_enc.hpp) for one limb count; instantiated once per
// translation unit (padic_enc_kernels.hip: 72 limbs, padic_enc36_kernels.hip: 36 limbs) to keep the compile times parallel.
#pragma once
#include "geo_ops.hpp"
#include "kernels_padic_enc.hpp"

namespace pai {

template <int NL, int U>
struct EncLaunch {
    static constexpr int BYTES2 = 2 * NL * BLOCK_THREADS * 4 + 2 * NL * 4;      // digit pair per lane + modulus copies
    static void fb_table(hipStream_t s, const MontCtx* nctx, const uint32_t* nm1, const uint32_t* hs_dig, const uint32_t* one_dig,
                         uint32_t* table, int J, int wb, const FbBases& fb) {
        constexpr int bytes = 3 * NL * 64 * 4 + 2 * NL * 4;
        (void)hipFuncSetAttribute((const void*)k_fb_table_padic<NL, U>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        hipLaunchKernelGGL((k_fb_table_padic<NL, U>), dim3((J + 63) / 64), dim3(64), bytes, s, nctx, nm1, hs_dig, one_dig,
                           reinterpret_cast<uint4*>(table), J, wb, fb.bases_plain, fb.base_words, fb.kdig, fb.nd);
    }
    static void fb_expand(hipStream_t s, int grid, const MontCtx* nctx, const uint32_t* nm1, const uint32_t* S, uint32_t* T, int J,
                          int h, uint32_t* mscratch) {
        (void)hipFuncSetAttribute((const void*)k_fb_expand_padic<NL, U>, hipFuncAttributeMaxDynamicSharedMemorySize, BYTES2);
        hipLaunchKernelGGL((k_fb_expand_padic<NL, U>), dim3(grid), dim3(BLOCK_THREADS), BYTES2, s, nctx, nm1,
                           reinterpret_cast<const uint4*>(S), reinterpret_cast<uint4*>(T), J, h, reinterpret_cast<uint4*>(mscratch));
    }
    static void encrypt(hipStream_t s, int grid, const EncPadicParams& P, const uint32_t* m, const uint32_t* r, const uint32_t* ct_in,
                        uint32_t* ct_out, int n, int mode) {
        // one instantiation per (apply_obfuscator?, g-factored table?): the table format is fixed when the table is built
        auto go = [&](auto kernel) {
            (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, BYTES2);
            hipLaunchKernelGGL(kernel, dim3(grid), dim3(BLOCK_THREADS), BYTES2, s, P, m, r, ct_in, ct_out, n, mode);
        };
        const bool gf = PAI_ENC_GFORM_OK && P.fb_gform != 0;
        if (mode == 2) {
            if (gf) go(k_encrypt_padic<NL, U, true, PAI_ENC_GFORM_OK>);
            else go(k_encrypt_padic<NL, U, true, false>);
        } else {
            if (gf) go(k_encrypt_padic<NL, U, false, PAI_ENC_GFORM_OK>);
            else go(k_encrypt_padic<NL, U, false, false>);
        }
    }
    // g-factoring passes over a finished table (kernels_padic_enc.hpp: k_fb_g_prefix / k_fb_g_finish)
    static void g_prefix(hipStream_t s, int grid, const MontCtx* nctx, const uint32_t* table, size_t count, int K, uint32_t* pref,
                         uint32_t* tot, int tw, uint32_t* mscratch) {
        (void)hipFuncSetAttribute((const void*)k_fb_g_prefix<NL, U>, hipFuncAttributeMaxDynamicSharedMemorySize, BYTES2);
        hipLaunchKernelGGL((k_fb_g_prefix<NL, U>), dim3(grid), dim3(BLOCK_THREADS), BYTES2, s, nctx, reinterpret_cast<const uint4*>(table),
                           count, K, reinterpret_cast<uint4*>(pref), tot, tw, reinterpret_cast<uint4*>(mscratch));
    }
    static void g_finish(hipStream_t s, int grid, const MontCtx* nctx, uint32_t* table, size_t count, int K, const uint32_t* pref,
                         const uint32_t* inv, int tw, uint32_t* mscratch) {
        (void)hipFuncSetAttribute((const void*)k_fb_g_finish<NL, U>, hipFuncAttributeMaxDynamicSharedMemorySize, BYTES2);
        hipLaunchKernelGGL((k_fb_g_finish<NL, U>), dim3(grid), dim3(BLOCK_THREADS), BYTES2, s, nctx, reinterpret_cast<uint4*>(table), count,
                           K, reinterpret_cast<const uint4*>(pref), inv, tw, reinterpret_cast<uint4*>(mscratch));
    }
    static void ctmul(hipStream_t s, int grid, const CtMulPadicParams& P, const uint32_t* ct, const uint32_t* e, uint32_t* out, int n) {
        (void)hipFuncSetAttribute((const void*)k_ctmul_padic<NL, U>, hipFuncAttributeMaxDynamicSharedMemorySize, BYTES2);
        hipLaunchKernelGGL((k_ctmul_padic<NL, U>), dim3(grid), dim3(BLOCK_THREADS), BYTES2, s, P, ct, e, out, n);
    }
    static void mexp_table(hipStream_t s, int grid, const MexpPadicParams& P, const uint32_t* ct, const uint32_t* ct_inv, int nlanes) {
        (void)hipFuncSetAttribute((const void*)k_mexp_table_padic<NL, U>, hipFuncAttributeMaxDynamicSharedMemorySize, BYTES2);
        hipLaunchKernelGGL((k_mexp_table_padic<NL, U>), dim3(grid), dim3(BLOCK_THREADS), BYTES2, s, P, ct, ct_inv, nlanes);
    }
    static void mexp(hipStream_t s, int grid, const MexpPadicParams& P, const uint32_t* e, const uint8_t* sign, uint32_t* out, int nlanes) {
        (void)hipFuncSetAttribute((const void*)k_mexp_padic<NL, U>, hipFuncAttributeMaxDynamicSharedMemorySize, BYTES2);
        hipLaunchKernelGGL((k_mexp_padic<NL, U>), dim3(grid), dim3(BLOCK_THREADS), BYTES2, s, P, e, sign, out, nlanes);
    }
    static void pow(hipStream_t s, int grid, const PowPadicParams& P, const uint32_t* base, uint32_t* out, int n) {
        (void)hipFuncSetAttribute((const void*)k_pow_padic<NL, U>, hipFuncAttributeMaxDynamicSharedMemorySize, BYTES2);
        hipLaunchKernelGGL((k_pow_padic<NL, U>), dim3(grid), dim3(BLOCK_THREADS), BYTES2, s, P, base, out, n);
    }
};

// 36-limb instantiations (padic_enc36_kernels.hip)
void enc36_fb_table(hipStream_t s, const MontCtx* nctx, const uint32_t* nm1, const uint32_t* hs_dig, const uint32_t* one_dig,
                    uint32_t* table, int J, int wb, const FbBases& fb);
void enc36_fb_expand(hipStream_t s, int grid, const MontCtx* nctx, const uint32_t* nm1, const uint32_t* S, uint32_t* T, int J, int h,
                     uint32_t* mscratch);
void enc36_encrypt(hipStream_t s, int grid, const EncPadicParams& P, const uint32_t* m, const uint32_t* r, const uint32_t* ct_in,
                   uint32_t* ct_out, int n, int mode);
void enc36_ctmul(hipStream_t s, int grid, const CtMulPadicParams& P, const uint32_t* ct, const uint32_t* e, uint32_t* out, int n);
void enc36_g_prefix(hipStream_t s, int grid, const MontCtx* nctx, const uint32_t* table, size_t count, int K, uint32_t* pref, uint32_t* tot,
                    int tw, uint32_t* mscratch);
void enc36_g_finish(hipStream_t s, int grid, const MontCtx* nctx, uint32_t* table, size_t count, int K, const uint32_t* pref,
                    const uint32_t* inv, int tw, uint32_t* mscratch);
void enc36_pow(hipStream_t s, int grid, const PowPadicParams& P, const uint32_t* base, uint32_t* out, int n);
void enc36_mexp_table(hipStream_t s, int grid, const MexpPadicParams& P, const uint32_t* ct, const uint32_t* ct_inv, int nlanes);
void enc36_mexp(hipStream_t s, int grid, const MexpPadicParams& P, const uint32_t* e, const uint8_t* sign, uint32_t* out, int nlanes);

}  // namespace pai
