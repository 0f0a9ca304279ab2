// Instantiations of the base-n digit-pair kernels (kernels_padic_enc.hpp): fixed-base table construction, raw / DJN
// encryption and ciphertext * plaintext on 72 limbs.  Own translation unit (see padic_dec_kernels.hip).
#include "geo_ops.hpp"
#include "kernels_padic_enc.hpp"

namespace pai {

#ifndef PAI_ENC_U
#define PAI_ENC_U 8
#endif
constexpr int ENC_U = PAI_ENC_U;     // rows per block of the 72-limb products

// ---- digit engine with base n for encryption (kernels_padic_enc.hpp): 1400..2048-bit n, 72 limbs -------
int padic_enc_nl_for_n_bits(int bits) { return (bits >= 1400 && RB * 72 >= bits + 20) ? 72 : 0; }
bool launch_fb_table_padic(int nl, hipStream_t s, const MontCtx* nctx, const uint32_t* nm1, const uint32_t* hs_dig,
                           const uint32_t* one_dig, uint32_t* table, int J, int wb) {
    if (nl != 72) return false;
    constexpr int bytes = 3 * 72 * 64 * 4 + 2 * 72 * 4;
    (void)hipFuncSetAttribute((const void*)k_fb_table_padic<72, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    hipLaunchKernelGGL((k_fb_table_padic<72, 8>), dim3((J + 63) / 64), dim3(64), bytes, s, nctx, nm1, hs_dig, one_dig,
                       reinterpret_cast<uint4*>(table), J, wb);
    return true;
}
bool launch_fb_expand_padic(int nl, hipStream_t s, int grid, const MontCtx* nctx, const uint32_t* nm1, const uint32_t* S,
                            uint32_t* T, int J, int h, uint32_t* mscratch) {
    if (nl != 72) return false;
    constexpr int bytes = 2 * 72 * BLOCK_THREADS * 4 + 2 * 72 * 4;
    (void)hipFuncSetAttribute((const void*)k_fb_expand_padic<72, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    hipLaunchKernelGGL((k_fb_expand_padic<72, 8>), dim3(grid), dim3(BLOCK_THREADS), bytes, s, nctx, nm1,
                       reinterpret_cast<const uint4*>(S), reinterpret_cast<uint4*>(T), J, h, reinterpret_cast<uint4*>(mscratch));
    return true;
}
bool launch_encrypt_padic(int nl, hipStream_t s, int grid, const EncPadicParams& P, const uint32_t* m, const uint32_t* r,
                          const uint32_t* ct_in, uint32_t* ct_out, int n, int mode) {
    if (nl != 72) return false;
    constexpr int bytes = 2 * 72 * BLOCK_THREADS * 4 + 2 * 72 * 4;
    if (mode == 2) {
        (void)hipFuncSetAttribute((const void*)k_encrypt_padic<72, ENC_U, true>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        hipLaunchKernelGGL((k_encrypt_padic<72, ENC_U, true>), dim3(grid), dim3(BLOCK_THREADS), bytes, s, P, m, r, ct_in, ct_out, n, mode);
    } else {
        (void)hipFuncSetAttribute((const void*)k_encrypt_padic<72, ENC_U, false>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        hipLaunchKernelGGL((k_encrypt_padic<72, ENC_U, false>), dim3(grid), dim3(BLOCK_THREADS), bytes, s, P, m, r, ct_in, ct_out, n, mode);
    }
    return true;
}

size_t ctmul_padic_table_words(int nl, int wbits, size_t blocks) { return ((size_t)1 << wbits) * 2 * nl * blocks * BLOCK_THREADS; }
bool launch_ctmul_padic(int nl, hipStream_t s, int grid, const CtMulPadicParams& P, const uint32_t* ct, const uint32_t* e,
                        uint32_t* out, int n) {
    if (nl != 72) return false;
    constexpr int bytes = 2 * 72 * BLOCK_THREADS * 4 + 2 * 72 * 4;
    (void)hipFuncSetAttribute((const void*)k_ctmul_padic<72, ENC_U>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    hipLaunchKernelGGL((k_ctmul_padic<72, ENC_U>), dim3(grid), dim3(BLOCK_THREADS), bytes, s, P, ct, e, out, n);
    return true;
}

bool launch_pow_padic(int nl, hipStream_t s, int grid, const PowPadicParams& P, const uint32_t* base, uint32_t* out, int n) {
    if (nl != 72) return false;
    constexpr int bytes = 2 * 72 * BLOCK_THREADS * 4 + 2 * 72 * 4;
    (void)hipFuncSetAttribute((const void*)k_pow_padic<72, ENC_U>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    hipLaunchKernelGGL((k_pow_padic<72, ENC_U>), dim3(grid), dim3(BLOCK_THREADS), bytes, s, P, base, out, n);
    return true;
}

}  // namespace pai
