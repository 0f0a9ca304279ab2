// Kernel instantiations for the geometry: 36 radix-2^29 limbs per lane x 8 lanes per integer.
#include "geo_inst.hpp"
#ifndef PAI_U_36X8
#define PAI_U_36X8 6
#endif
namespace pai { const GeoOps* geo_ops_36x8() { return GeoInst<Geo<36, 8, PAI_U_36X8, false>>::ops(); } }
