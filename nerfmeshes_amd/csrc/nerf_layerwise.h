// The layer-wise network path (nerf_layerwise.hip): handle-side description and entry points.
#pragma once
#include <vector>

#include "nm_internal.h"

namespace nm {

constexpr int LW_MAX_FREQ = 32;     // encoding functions per input the layer-wise path takes (2^31 is past fp32's integer range anyway)

// one torch.nn.Linear (out x in) inside the handle's blob (offsets in floats): W^T (in x out) for the forward products,
// W (out x in) for the delta chain, the bias
struct LwLinear { size_t wt, w, b; int out, in; };

struct LwNet {
    LwLinear layer1, xyz[32], feat, dir, alpha, rgb;     // rgb: fc_rgb (3 x H/2), or rows 0..2 of fc_out (3 x H) without view directions
    int L, H, H2, dx, dd, flat;
    uint32_t skip_mask;                                  // bit i: layers_xyz[i] consumes cat(hidden, xyz encoding)
    int fx, fd, inc_x, inc_d;
    float bands_x[LW_MAX_FREQ], bands_d[LW_MAX_FREQ];
    float* ws;                                           // activation planes of one batch (grow-only)
    size_t ws_floats;
};

int layerwise_forward(nm_mlp* m, const MlpArgs& args, int density_only, hipStream_t stream);
int layerwise_forward_train(nm_mlp* m, const MlpArgs& args, const nm_mlp_tape* tape, hipStream_t stream);
int layerwise_backward(nm_mlp* m, int64_t n, const nm_mlp_tape* tape, const float* d_radiance, const float* d_grad_radiance,
                       const nm_mlp_deltas* deltas, hipStream_t stream);
void layerwise_destroy(nm_mlp* m);

}  // namespace nm
