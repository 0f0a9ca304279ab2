// Kernel instantiations for the geometry: 28 radix-2^29 limbs per lane x 4 lanes per integer.
#include "geo_inst.hpp"
namespace pai { const GeoOps* geo_ops_28x4() { return GeoInst<Geo<28, 4, 4, false>>::ops(); } }
