// Kernel instantiations for the LATENCY geometry: 3 radix-2^29 limbs per lane x 64 lanes per integer — an integer is
// spread over the whole wavefront (quotient digits broadcast through SGPRs, limb hand-over by wave_shl DPP), which
// divides the latency of one Montgomery product by ~64 at a fraction of the multiplier efficiency: the engine
// behind small batches (the reference's own benchmark sizes are 16 and 64 elements).
#include "geo_inst.hpp"
namespace pai { const GeoOps* geo_ops_3x64() { return GeoInst<Geo<3, 64, 3, false>>::ops(); } }
