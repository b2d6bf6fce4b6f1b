// The upper half (30 / 32 tiles) of nerf_mlp_generic_s_long.hip's instantiations: a translation unit of its own, for build time only.
#define NM_GENERIC_LONG 1
#define NM_GENERIC_UPPER 1
#include "nerf_mlp_generic_s.hip"
