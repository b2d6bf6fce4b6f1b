// The split width classes of nerf_mlp_generic_s.hip once more, with two-part encoding stages (16 -- 31 functions): see there.
#define NM_GENERIC_LONG 1
#include "nerf_mlp_generic_s.hip"
