// Instantiations of the digit-pair decrypt kernel (kernels_padic.hpp).  Own translation unit: the kernels are large
// (seconds of compile time each) and their scheduler flags can be varied independently (build.py).
#include "geo_ops.hpp"
#include "kernels_padic.hpp"

namespace pai {

// ---- p-adic digit engine (kernels_padic.hpp): primes of 400..676 bits on 24 limbs and up to 1024 bits on 36 limbs
// (12-row blocks, everything in LDS), 1025..1604 bits on 56 limbs and up to 2068 bits on 72 limbs (8-row blocks, quotient digits in scratch)
int padic_nl_for_prime_bits(int bits) {
    if (bits < 400) return 0;
    if (RB * 24 >= bits + 20) return 24;
    if (RB * 36 >= bits + 20) return 36;
    if (RB * 56 >= bits + 20) return 56;
    if (RB * 72 >= bits + 20) return 72;
    return 0;
}
size_t padic_table_words(int nl, size_t blocks) { return (size_t)(PADIC_TBL_ENTRIES + 1) * 2 * nl * blocks * BLOCK_THREADS; }
int padic_blocks_per_cu(int) { return 1; }
size_t padic_scratch_words(int nl, size_t blocks) { return nl <= 36 ? 0 : (size_t)nl * blocks * BLOCK_THREADS; }
// rows per block: 12 at 24 / 36 limbs (4-row blocks: 506.9 vs 477.8 ms per 2^20, profiles/r04/README.md), 8 at 56 / 72
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
        case 24: launch_padic<24, 12, PADIC_LDS_M>(s, gridx, P, ct, u_out, n, table); return true;
        case 36: launch_padic<36, 12, PADIC_LDS_M>(s, gridx, P, ct, u_out, n, table); return true;
        case 56: launch_padic<56, 8, PADIC_WBUF>(s, gridx, P, ct, u_out, n, table); return true;
        case 72: launch_padic<72, 8, PADIC_WBUF>(s, gridx, P, ct, u_out, n, table); return true;
        default: return false;
    }
}

}  // namespace pai
