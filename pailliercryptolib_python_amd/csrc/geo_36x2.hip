// Kernel instantiations for the geometry: 36 radix-2^29 limbs per lane x 2 lanes per integer.
#include "geo_inst.hpp"
namespace pai { const GeoOps* geo_ops_36x2() { return GeoInst<Geo<36, 2, 6, false>>::ops(); } }
