// The ONE-WAVE-PER-SIMD kernels of the width classes NT = 26, 28, 30, 32 (hidden_size 385 -- 512): what served these classes until
// round 5, kept in the ablation library only (-DNM_ABLATIONS, NM_MLP_VARIANT=310) as the A side of tests/tools/bench_split.py.  The
// product library runs these classes on the split kernels of mlp_device_gs.h (nerf_mlp_generic_s.hip).
#include <vector>

#include "nm_internal.h"
#include "mlp_device_g.h"

namespace nm {

#ifdef NM_ABLATIONS
template <int NT>
static MlpPlan generic_plan() {
    constexpr int NW = 4, KCH = 4;            // one wave per SIMD on the 512-register budget; ring slots of at most 32 KiB
    constexpr int SLOT = KCH * ((NT + 3) / 4) * 1024;
    return MlpPlan{16 * NT, -1, -1, NW, KCH, 310, 2 * SLOT, true, &mlp_kernel_g<NT, NW, KCH>, NW * 16, 1,
                   &mlp_kernel_g<NT, NW, KCH>, NT, &mlp_kernel_g<NT, NW, KCH, true>, &mlp_backward_kernel_g<NT, NW, KCH>};
}
#endif

void generic_plans_e(std::vector<MlpPlan>& out) {
#ifdef NM_ABLATIONS
    out.push_back(generic_plan<26>());
    out.push_back(generic_plan<28>());
    out.push_back(generic_plan<30>());
    out.push_back(generic_plan<32>());
#else
    (void)out;
#endif
}

}  // namespace nm
