// Instantiations of the wide-engine kernels (kernels_wide.hpp): CRT-decrypt stage A for s^2 of up to
// 1158 bits (40 limbs) and up to 2086 bits (72 limbs).
#include "geo_ops.hpp"
#include "kernels_padic.hpp"
#include "kernels_padic_enc.hpp"
#include "kernels_wide.hpp"

namespace pai {

template <int NL>
static void launch_a(hipStream_t s, int gridx, const DecAParams& P, const uint32_t* ct, uint32_t* u_out, int n, uint32_t* table) {
    constexpr int bytes = NL * BLOCK_THREADS * 4;
    (void)hipFuncSetAttribute((const void*)k_dec_a_wide<NL, MODEXP_WINDOW>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    hipLaunchKernelGGL((k_dec_a_wide<NL, MODEXP_WINDOW>), dim3(gridx, 2), dim3(BLOCK_THREADS), bytes, s, P, ct, u_out, n,
                       reinterpret_cast<uint4*>(table));
}

int wide_nl_for_bits(int bits) {
    if (RB * 40 >= bits + 2) return 40;
    if (RB * 72 >= bits + 2) return 72;
    return 0;
}

size_t wide_table_words(int nl, size_t blocks) { return (size_t)(1u << MODEXP_WINDOW) * nl * blocks * BLOCK_THREADS; }

bool launch_dec_a_wide(int nl, hipStream_t s, int gridx, const DecAParams& P, const uint32_t* ct, uint32_t* u_out, int n,
                       uint32_t* table) {
    switch (nl) {
        case 40: launch_a<40>(s, gridx, P, ct, u_out, n, table); return true;
        case 72: launch_a<72>(s, gridx, P, ct, u_out, n, table); return true;
        default: return false;
    }
}

// ---- p-adic digit engine (kernels_padic.hpp): primes of 700..1024 bits on 36 limbs (12-row blocks, everything
// in LDS), 1025..1604 bits on 56 limbs and up to 2068 bits on 72 limbs (8-row blocks, quotient digits in scratch)
int padic_nl_for_prime_bits(int bits) {
    if (bits < 700) return 0;
    if (RB * 36 >= bits + 20) return 36;
    if (RB * 56 >= bits + 20) return 56;
    if (RB * 72 >= bits + 20) return 72;
    return 0;
}
size_t padic_table_words(int nl, size_t blocks) { return (size_t)(PADIC_TBL_ENTRIES + 1) * 2 * nl * blocks * BLOCK_THREADS; }
size_t padic_scratch_words(int nl, size_t blocks) { return nl == 36 ? 0 : 2 * (size_t)nl * blocks * BLOCK_THREADS; }
template <int NL, int U, int MODE>
static void launch_padic(hipStream_t s, int gridx, const DecPadicParams& P, const uint32_t* ct, uint32_t* u_out, int n, uint32_t* table) {
    constexpr int bytes = (MODE == PADIC_LDS_M ? 3 : 2) * NL * BLOCK_THREADS * 4 + 2 * NL * 4;
    (void)hipFuncSetAttribute((const void*)k_dec_a_padic<NL, U, MODEXP_WINDOW, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    hipLaunchKernelGGL((k_dec_a_padic<NL, U, MODEXP_WINDOW, MODE>), dim3(gridx, 2), dim3(BLOCK_THREADS), bytes, s, P, ct, u_out, n,
                       reinterpret_cast<uint4*>(table));
}
bool launch_dec_a_padic(int nl, hipStream_t s, int gridx, const DecPadicParams& P, const uint32_t* ct,
                        uint32_t* u_out, int n, uint32_t* table) {
    switch (nl) {
        case 36: launch_padic<36, 12, PADIC_LDS_M>(s, gridx, P, ct, u_out, n, table); return true;
        case 56: launch_padic<56, 8, PADIC_WBUF>(s, gridx, P, ct, u_out, n, table); return true;
        case 72: launch_padic<72, 8, PADIC_WBUF>(s, gridx, P, ct, u_out, n, table); return true;
        default: return false;
    }
}

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
                          uint32_t* ct_out, int n, int mode) {
    if (nl != 72) return false;
    constexpr int bytes = 2 * 72 * BLOCK_THREADS * 4 + 2 * 72 * 4;
    (void)hipFuncSetAttribute((const void*)k_encrypt_padic<72, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    hipLaunchKernelGGL((k_encrypt_padic<72, 8>), dim3(grid), dim3(BLOCK_THREADS), bytes, s, P, m, r, nullptr, ct_out, n, mode);
    return true;
}

size_t ctmul_padic_table_words(int nl, int wbits, size_t blocks) { return ((size_t)1 << wbits) * 2 * nl * blocks * BLOCK_THREADS; }
bool launch_ctmul_padic(int nl, hipStream_t s, int grid, const CtMulPadicParams& P, const uint32_t* ct, const uint32_t* e,
                        uint32_t* out, int n) {
    if (nl != 72) return false;
    constexpr int bytes = 2 * 72 * BLOCK_THREADS * 4 + 2 * 72 * 4;
    (void)hipFuncSetAttribute((const void*)k_ctmul_padic<72, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    hipLaunchKernelGGL((k_ctmul_padic<72, 8>), dim3(grid), dim3(BLOCK_THREADS), bytes, s, P, ct, e, out, n);
    return true;
}

}  // namespace pai
