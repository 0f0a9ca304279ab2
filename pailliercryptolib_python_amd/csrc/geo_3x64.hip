// Kernel instantiations for the LATENCY geometry: 3 radix-2^29 limbs per lane x 64 lanes per integer — an integer is
// spread over the whole wavefront (quotient digits broadcast through SGPRs, limb hand-over by wave_shl DPP), which
// divides the latency of one Montgomery product by ~64 at a fraction of the multiplier efficiency: the engine
// behind small batches (the reference's own benchmark sizes are 16 and 64 elements).
#include "geo_inst.hpp"
#include "kernels_declat.hpp"
namespace pai {
const GeoOps* geo_ops_3x64() { return GeoInst<Geo<3, 64, 3, false>>::ops(); }
// stage A of the smallest decryptions on digit pairs, four waves per (ciphertext, prime) (kernels_declat.hpp); chain_limbs:
// the limbs per lane the chain's contexts were built for (1: s' of at most 64 limbs, else 2)
template <class GC>
static void launch_pp(hipStream_t s, int n, int gy, bool var, const DecPPParams& P, const uint32_t* ct, uint32_t* out) {
    using GP = Geo<3, 64, 3, false, true>;
    constexpr int bytes = PPLds<GP, GC>::BYTES;
    if (var) {
        (void)hipFuncSetAttribute((const void*)k_ctmul_pp<GP, GC>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        hipLaunchKernelGGL((k_ctmul_pp<GP, GC>), dim3(n, gy), dim3(BLOCK_THREADS), bytes, s, P, ct, out, n);
    } else {
        (void)hipFuncSetAttribute((const void*)k_dec_a_pp<GP, GC>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        hipLaunchKernelGGL((k_dec_a_pp<GP, GC>), dim3(n, gy), dim3(BLOCK_THREADS), bytes, s, P, ct, out, n);
    }
}
void launch_dec_a_pp(hipStream_t s, int n, const DecPPParams& P, const uint32_t* ct, uint32_t* u_out, int chain_limbs) {
    if (chain_limbs == 1) launch_pp<Geo<1, 64, 1, false, true>>(s, n, 2, false, P, ct, u_out);
    else launch_pp<Geo<2, 64, 1, false, true>>(s, n, 2, false, P, ct, u_out);
}
// ct * pt of the smallest batches on the same four-wave pipeline, one workgroup per ciphertext (pp_chain<G, GC, true>)
void launch_ctmul_pp(hipStream_t s, int n, const DecPPParams& P, const uint32_t* ct, uint32_t* out, int chain_limbs) {
    if (chain_limbs == 1) launch_pp<Geo<1, 64, 1, false, true>>(s, n, 1, true, P, ct, out);
    else launch_pp<Geo<2, 64, 1, false, true>>(s, n, 1, true, P, ct, out);
}
}  // namespace pai
