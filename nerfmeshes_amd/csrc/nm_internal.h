// Internal declarations shared by the HIP translation units of libnerfmeshes_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/nerfmeshes_hip.h"

namespace nm {

void set_error(const std::string& msg);

#define NM_HIP_CHECK(expr)                                                                   \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) {                                                              \
            ::nm::set_error(std::string(#expr) + ": " + hipGetErrorString(_e));              \
            return 1;                                                                        \
        }                                                                                    \
    } while (0)

#define NM_REQUIRE(cond, msg)                                                                \
    do {                                                                                     \
        if (!(cond)) {                                                                       \
            ::nm::set_error(std::string(msg) + " [" #cond "]");                              \
            return 2;                                                                        \
        }                                                                                    \
    } while (0)

// ---- fused MLP ------------------------------------------------------------------------------
constexpr int KC = 8;          // k-steps (of 4 input features) per LDS weight chunk
constexpr int MAX_FREQ_XYZ = 16;
constexpr int MAX_FREQ_DIR = 8;

enum MlpMode : int { MODE_POINTS = 0, MODE_RAYS = 1, MODE_GRID = 2 };

// Kernel arguments (passed by value).
struct MlpArgs {
    const char* wstream;     // packed MFMA A-operand stream, stage after stage
    const float* bias;       // layer1 | layers_xyz.* | fc_feat | layers_dir.0 | fc_alpha.b | fc_rgb.b[3]  (concatenated)
    const float* walpha;     // [4 lane groups][H/4 k-steps]
    const float* wrgb;       // [3][4][H/8]
    float bands_xyz[MAX_FREQ_XYZ];
    float bands_dir[MAX_FREQ_DIR];
    uint32_t skip_mask;      // bit i set: layers_xyz[i] consumes cat(hidden, xyz_enc)
    int32_t mode;
    // inputs: POINTS: a=points b=dirs | RAYS: a=origins b=dirs c=t | GRID: a,b,c = axis values
    const float* a;
    const float* b;
    const float* c;
    int64_t n;               // samples to evaluate
    int64_t first;           // GRID: flat index of the first point
    int32_t samples;         // RAYS: samples per ray
    int32_t origins_per_ray; // RAYS
    int32_t n1, n2;          // GRID: sizes of axis 1 and 2
    float* out;              // (n,4) or (n,) when density only
    // training tape (nm_mlp_forward_train only; see nm_mlp_tape in the public header)
    float* tape_h;           // (L, n, H): layer1 output, then the post-ReLU output of every layers_xyz[i]
    float* tape_feat;        // (n, H)  relu(fc_feat(x))
    float* tape_v;           // (n, H/2)
    uint64_t* mask_h;        // (L, tiles, 64): layers_xyz[0..L-2] then fc_feat; per lane, bit 4*tile+reg = activation > 0
    uint64_t* mask_v;        // (tiles, 64)
    int64_t tiles;           // ceil(n / 16)
};

// Kernel arguments of the backward (delta propagation) kernel.
struct MlpBwdArgs {
    const char* wstream;     // transposed A-operand stream: layers_dir.0^T | fc_feat^T | layers_xyz[L-2..0]^T
    const float* walpha;
    const float* wrgb;
    const float* radiance;   // (n,4) forward output (sigmoid(rgb), sigma)
    const float* grad_out;   // (n,4) dL/d(radiance)
    const uint64_t* mask_h;
    const uint64_t* mask_v;
    int64_t n, tiles;
    float* d_h;              // (L, n, H) gradient at layer1's output, then at every layers_xyz[i] pre-activation
    float* d_feat;           // (n, H)
    float* d_v;              // (n, H/2)
    float* d_last;           // (n, 4): pre-sigmoid rgb gradient, sigma gradient
};

// Flat addressing of the trainable tensors (index maps of the packed blob: tensor id << 24 | element)
enum TensorId : int {
    T_L1W = 0, T_L1B = 1, T_XYZ0 = 2 /* + 2i weight, + 2i + 1 bias */, T_FEATW = 66, T_FEATB, T_ALPHAW, T_ALPHAB,
    T_DIRW, T_DIRB, T_RGBW, T_RGBB, T_COUNT
};
struct WeightPtrs { const float* p[T_COUNT]; };

struct MlpPlan;  // host-side description of one template instantiation

}  // namespace nm

// The opaque handle of the C ABI.
struct nm_mlp {
    nm_mlp_desc desc;
    int device;
    const nm::MlpPlan* plan;
    void* d_blob;            // one device allocation holding stream + bias + walpha + wrgb
    size_t blob_bytes;
    nm::MlpArgs base;        // weight-related fields filled in
    nm::MlpBwdArgs bwd;      // likewise for the backward kernel
    int32_t* d_index;        // source of every float of the blob (tensor id << 24 | element, -1 = 0.0f)
    size_t blob_floats;
    int64_t flops_full, flops_density;
    int num_cus;
};
