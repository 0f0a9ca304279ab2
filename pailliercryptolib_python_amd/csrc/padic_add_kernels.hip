// ct + ct by true division on the one-element-per-lane digit engine (kernels_ctadd_div.hpp), 72 limbs: the 2048-bit keys.
#include "geo_ops.hpp"
#include "kernels_ctadd_div.hpp"

namespace pai {

size_t ctadd_div_scratch_bytes(int nl, size_t blocks) { return (size_t)6 * (nl / 4) * 16 * blocks * BLOCK_THREADS; }

bool launch_ctadd_div(int nl, hipStream_t s, int grid, const CtAddDivParams& P, const uint32_t* a, const uint32_t* b, uint32_t* out, int n) {
    if (nl != 72) return false;
    constexpr int NL = 72, U = 12;
    constexpr int bytes = 2 * NL * BLOCK_THREADS * 4 + 2 * NL * 4;          // digit pair per lane + n and mu
    (void)hipFuncSetAttribute((const void*)k_ctadd_div<NL, U>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    hipLaunchKernelGGL((k_ctadd_div<NL, U>), dim3(grid), dim3(BLOCK_THREADS), bytes, s, P, a, b, out, n);
    return true;
}

}  // namespace pai
