// Generic-shape instantiations of the fused MLP, the classes of 26 -- 32 tiles (hidden_size 385 -- 512) with a layer's output tiles
// split over a pair of waves (mlp_device_gs.h): 8-wave workgroups, two waves per SIMD.
#include <vector>

#include "nm_internal.h"
#include "mlp_device_gs.h"

// compiled twice: as it is, and from nerf_mlp_generic_s_long.hip with NM_GENERIC_LONG defined (two-part encoding stages, 16 -- 31
// functions: plans of variant G_LONG_VARIANT)
#if defined(NM_GENERIC_LONG) && defined(NM_GENERIC_UPPER)
#define NM_PLANS_FN generic_plans_s_long_upper      // (the long-encoding unit was the build's critical path: 172 s; halved)
constexpr bool kLong = true;
#elif defined(NM_GENERIC_LONG)
#define NM_PLANS_FN generic_plans_s_long
constexpr bool kLong = true;
#else
#define NM_PLANS_FN generic_plans_s
constexpr bool kLong = false;
#endif

namespace nm {

template <int NT>
static MlpPlan split_plan() {
    constexpr int KCH = 4;                    // one input tile per chunk (the exchange schedule of mlp_device_gs.h)
    constexpr int SLOT = KCH * ((NT + 3) / 4) * 1024;
    return MlpPlan{16 * NT, -1, -1, 2 * GS_PAIRS, KCH, kLong ? G_LONG_VARIANT : 0, 2 * SLOT + GS_EXTRA_BYTES, true, &mlp_kernel_gs<NT, KCH, false, kLong>, GS_PAIRS * 16, 1,
                   &mlp_kernel_gs<NT, KCH, false, kLong>, NT, &mlp_kernel_gs<NT, KCH, true, kLong>, &mlp_backward_kernel_gs<NT, KCH>};
}

void NM_PLANS_FN(std::vector<MlpPlan>& out) {
#if !defined(NM_GENERIC_UPPER)
    out.push_back(split_plan<26>());
    out.push_back(split_plan<28>());
#endif
#if !defined(NM_GENERIC_LONG) || defined(NM_GENERIC_UPPER)
    out.push_back(split_plan<30>());
    out.push_back(split_plan<32>());
#endif
}

}  // namespace nm
