// Generic-shape instantiations of the fused MLP (mlp_device_g.h), part a: width classes NT = 1, 2, 3, 4, 5, 6, 7 (hidden_size <= 16 NT).
// One translation unit per group of classes so that the build compiles them side by side (nerfmeshes_amd/build.py).
#include <vector>

#include "nm_internal.h"
#include "mlp_device_g.h"

namespace nm {

template <int NT>
static MlpPlan generic_plan() {
    static_assert(NT <= 24, "wider classes: nerf_mlp_generic_s.hip");
    constexpr int NW = 8, KCH = 8;            // two waves per SIMD; ring slots of at most 48 KiB
    constexpr int SLOT = KCH * ((NT + 3) / 4) * 1024;
    return MlpPlan{16 * NT, -1, -1, NW, KCH, 0, 2 * SLOT, true, &mlp_kernel_g<NT, NW, KCH>, NW * 16, 1,
                   &mlp_kernel_g<NT, NW, KCH>, NT, &mlp_kernel_g<NT, NW, KCH, true>, &mlp_backward_kernel_g<NT, NW, KCH>};
}

void generic_plans_a(std::vector<MlpPlan>& out) {
    out.push_back(generic_plan<1>());
    out.push_back(generic_plan<2>());
    out.push_back(generic_plan<3>());
    out.push_back(generic_plan<4>());
    out.push_back(generic_plan<5>());
    out.push_back(generic_plan<6>());
    out.push_back(generic_plan<7>());
}

}  // namespace nm
