// Kernel instantiations for the geometry: 36 radix-2^29 limbs per lane x 4 lanes per integer.
#include "geo_inst.hpp"
#ifndef PAI_U_36X4
#define PAI_U_36X4 6
#endif
namespace pai { const GeoOps* geo_ops_36x4() { return GeoInst<Geo<36, 4, PAI_U_36X4, false>>::ops(); } }
