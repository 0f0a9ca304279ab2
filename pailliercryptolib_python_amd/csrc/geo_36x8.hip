// Kernel instantiations for the geometry: 36 radix-2^29 limbs per lane x 8 lanes per integer.
#include "geo_inst.hpp"
namespace pai { const GeoOps* geo_ops_36x8() { return GeoInst<Geo<36, 8, 6, false>>::ops(); } }
