// Generic-shape instantiations of the fused MLP (mlp_device_g.h), part b: width classes NT = 8, 9, 10, 11, 12, 13 (hidden_size <= 16 NT).
// One translation unit per group of classes so that the build compiles them side by side (nerfmeshes_amd/build.py).
#include <vector>

#include "nm_internal.h"
#include "mlp_device_g.h"

// compiled twice: as it is, and from nerf_mlp_generic_b_long.hip with NM_GENERIC_LONG defined -- the instantiations whose encoding
// stages take two parts (16 -- 31 functions; enc_stages_g in mlp_device_g.h), registered as plans of variant G_LONG_VARIANT
#ifdef NM_GENERIC_LONG
#define NM_PLANS_FN generic_plans_b_long
constexpr bool kLong = true;
#else
#define NM_PLANS_FN generic_plans_b
constexpr bool kLong = false;
#endif

namespace nm {

template <int NT>
static MlpPlan generic_plan() {
    static_assert(NT <= 24, "wider classes: nerf_mlp_generic_s.hip");
    constexpr int NW = 8, KCH = 8;            // two waves per SIMD; ring slots of at most 48 KiB
    constexpr int SLOT = KCH * ((NT + 3) / 4) * 1024;
    return MlpPlan{16 * NT, -1, -1, NW, KCH, kLong ? G_LONG_VARIANT : 0, 2 * SLOT, true, &mlp_kernel_g<NT, NW, KCH, false, kLong>, NW * 16, 1,
                   &mlp_kernel_g<NT, NW, KCH, false, kLong>, NT, &mlp_kernel_g<NT, NW, KCH, true, kLong>, &mlp_backward_kernel_g<NT, NW, KCH>};
}

void NM_PLANS_FN(std::vector<MlpPlan>& out) {
    out.push_back(generic_plan<8>());
    out.push_back(generic_plan<9>());
    out.push_back(generic_plan<10>());
    out.push_back(generic_plan<11>());
    out.push_back(generic_plan<12>());
    out.push_back(generic_plan<13>());
}

}  // namespace nm
