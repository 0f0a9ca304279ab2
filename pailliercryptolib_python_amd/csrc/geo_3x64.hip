// Kernel instantiations for the LATENCY geometry: 3 radix-2^29 limbs per lane x 64 lanes per integer — an integer is
// spread over the whole wavefront (quotient digits broadcast through SGPRs, limb hand-over by wave_shl DPP), which
// divides the latency of one Montgomery product by ~64 at a fraction of the multiplier efficiency: the engine
// behind small batches (the reference's own benchmark sizes are 16 and 64 elements).
#include "geo_inst.hpp"
#include "kernels_declat.hpp"
namespace pai {
const GeoOps* geo_ops_3x64() { return GeoInst<Geo<3, 64, 3, false>>::ops(); }
// stage A of the smallest decryptions on digit pairs, four waves per (ciphertext, prime) (kernels_declat.hpp)
void launch_dec_a_pp(hipStream_t s, int n, const DecPPParams& P, const uint32_t* ct, uint32_t* u_out) {
    using GP = Geo<3, 64, 3, false, true>;
    constexpr int bytes = PPLds<GP>::BYTES;
    (void)hipFuncSetAttribute((const void*)k_dec_a_pp<GP>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    hipLaunchKernelGGL(k_dec_a_pp<GP>, dim3(n, 2), dim3(BLOCK_THREADS), bytes, s, P, ct, u_out, n);
}
// ct * pt of the smallest batches on the same four-wave pipeline, one workgroup per ciphertext (pp_chain<G, true>)
void launch_ctmul_pp(hipStream_t s, int n, const DecPPParams& P, const uint32_t* ct, uint32_t* out) {
    using GP = Geo<3, 64, 3, false, true>;
    constexpr int bytes = PPLds<GP>::BYTES;
    (void)hipFuncSetAttribute((const void*)k_ctmul_pp<GP>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    hipLaunchKernelGGL(k_ctmul_pp<GP>, dim3(n), dim3(BLOCK_THREADS), bytes, s, P, ct, out, n);
}
}  // namespace pai
