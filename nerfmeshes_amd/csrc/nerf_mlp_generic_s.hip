// Generic-shape instantiations of the fused MLP, the classes of 26 -- 32 tiles (hidden_size 385 -- 512) with a layer's output tiles
// split over a pair of waves (mlp_device_gs.h): 8-wave workgroups, two waves per SIMD.
#include <vector>

#include "nm_internal.h"
#include "mlp_device_gs.h"

namespace nm {

template <int NT>
static MlpPlan split_plan() {
    constexpr int KCH = 4;                    // one input tile per chunk (the exchange schedule of mlp_device_gs.h)
    constexpr int SLOT = KCH * ((NT + 3) / 4) * 1024;
    return MlpPlan{16 * NT, -1, -1, 2 * GS_PAIRS, KCH, 0, 2 * SLOT + GS_EXTRA_BYTES, true, &mlp_kernel_gs<NT, KCH>, GS_PAIRS * 16, 1,
                   &mlp_kernel_gs<NT, KCH>, NT, &mlp_kernel_gs<NT, KCH, true>, &mlp_backward_kernel_gs<NT, KCH>};
}

void generic_plans_s(std::vector<MlpPlan>& out) {
    out.push_back(split_plan<26>());
    out.push_back(split_plan<28>());
    out.push_back(split_plan<30>());
    out.push_back(split_plan<32>());
}

}  // namespace nm
