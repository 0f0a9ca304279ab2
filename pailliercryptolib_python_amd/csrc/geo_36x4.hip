// Kernel instantiations for the geometry: 36 radix-2^29 limbs per lane x 4 lanes per integer.
#include "geo_inst.hpp"
namespace pai { const GeoOps* geo_ops_36x4() { return GeoInst<Geo<36, 4, 6, false>>::ops(); } }
