// Kernel instantiations for the LATENCY geometry: 2 radix-2^29 limbs per lane x 64 lanes per integer (stage A of small-batch
// decryption on minus-one contexts: the fewest instructions per Montgomery row on a lone wave); otherwise as geo_3x64.hip:
// divides the latency of one Montgomery product by ~64 at a fraction of the multiplier efficiency: the engine
// behind small batches (the reference's own benchmark sizes are 16 and 64 elements).
#include "geo_inst.hpp"
namespace pai { const GeoOps* geo_ops_2x64() { return GeoInst<Geo<2, 64, 2, false>>::ops(); } }
