// Instantiations of the wave-parallel binary extended GCD (kernels_invert.hpp) for the ciphertext widths
// of 1024/2048/3072/4096-bit keys (64..256 words: 1..4 words per lane).
#include "geo_ops.hpp"
#include "kernels_invert.hpp"

namespace pai {

bool launch_inv_eea(hipStream_t s, int words, const uint32_t* mod, const uint32_t* a, uint32_t* out, int count,
                    int max_steps, int* fail) {
    if (count <= 0) return true;
    const int wpl = (words + 63) / 64;
    switch (wpl) {
        case 1: hipLaunchKernelGGL(k_inv_eea_wave<1>, dim3(count), dim3(64), 0, s, mod, a, out, words, max_steps, fail); return true;
        case 2: hipLaunchKernelGGL(k_inv_eea_wave<2>, dim3(count), dim3(64), 0, s, mod, a, out, words, max_steps, fail); return true;
        case 3: hipLaunchKernelGGL(k_inv_eea_wave<3>, dim3(count), dim3(64), 0, s, mod, a, out, words, max_steps, fail); return true;
        case 4: hipLaunchKernelGGL(k_inv_eea_wave<4>, dim3(count), dim3(64), 0, s, mod, a, out, words, max_steps, fail); return true;
        default: return false;
    }
}

}  // namespace pai
