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
    const float* bias;       // layer1 | layers_xyz.* | fc_feat | layers_dir.0   (concatenated)
    const float* walpha;     // [4 lane groups][H/4 k-steps]
    const float* wrgb;       // [3][4][H/8]
    float balpha;
    float brgb[3];
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
};

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
    int64_t flops_full, flops_density;
    int num_cus;
};
