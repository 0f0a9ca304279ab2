// Instantiations of the per-thread binary extended GCD (kernels_invert.hpp) for the ciphertext widths
// of 1024/2048/3072/4096-bit keys.
#include "geo_ops.hpp"

namespace pai {

template <int W>
static void launch_w(hipStream_t s, const uint32_t* mod, const uint32_t* a, uint32_t* out, int count, int max_steps, int* fail) {
    const int blocks = (count + 63) / 64;
    hipLaunchKernelGGL(k_inv_eea<W>, dim3(blocks), dim3(64), 0, s, mod, a, out, count, max_steps, fail);
}

bool launch_inv_eea(hipStream_t s, int words, const uint32_t* mod, const uint32_t* a, uint32_t* out, int count,
                    int max_steps, int* fail) {
    switch (words) {
        case 64: launch_w<64>(s, mod, a, out, count, max_steps, fail); return true;
        case 128: launch_w<128>(s, mod, a, out, count, max_steps, fail); return true;
        case 192: launch_w<192>(s, mod, a, out, count, max_steps, fail); return true;
        case 256: launch_w<256>(s, mod, a, out, count, max_steps, fail); return true;
        default: return false;
    }
}

}  // namespace pai
