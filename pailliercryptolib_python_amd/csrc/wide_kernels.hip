// Instantiations of the wide-engine kernels (kernels_wide.hpp): CRT-decrypt stage A for s^2 of up to
// 1158 bits (40 limbs) and up to 2086 bits (72 limbs).
#include "geo_ops.hpp"
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

}  // namespace pai
