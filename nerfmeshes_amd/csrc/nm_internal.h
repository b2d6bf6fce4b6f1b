// Internal declarations shared by the HIP translation units of libnerfmeshes_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
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
constexpr int MAX_FREQ_XYZ = 32;   // frequency bands a handle keeps per encoding (the fused kernels take up to 15 / 16 functions,
constexpr int MAX_FREQ_DIR = 32;   // the layer-wise path up to 32: nerf_layerwise.h)

enum MlpMode : int { MODE_POINTS = 0, MODE_RAYS = 1, MODE_GRID = 2, MODE_VIEW = 3 };

// Pinhole rays generated inside the consuming kernel from the camera pose (get_ray_bundle,
// /root/reference/src/nerf/nerf_helpers.py:226-277, optionally followed by ndc_rays, :280-307): ray r of a launch
// is pixel `first + r` (row-major).  The expressions are the ones ray_bundle_kernel / ndc_rays_kernel use, so a
// render from the pose is bit-identical to a render from a materialised ray buffer.
struct RayGen {
    float rot[9];            // c2w[:3,:3], row-major
    float origin[3];         // c2w[:3,3]
    int32_t height, width;
    float focal;
    int32_t enabled;
    int32_t ndc;             // apply ndc_rays(H, W, focal, near, o, d) to every generated ray
    float ndc_near, c_w, c_h, two_near, m_two_near;   // fp64 host constants of ndc_rays rounded to fp32, as torch does
    int64_t first;
};

__device__ __forceinline__ float nm_norm3(float x, float y, float z) {
    return sqrtf(fmaf(z, z, fmaf(y, y, x * x)));  // matches at::norm(p=2) on CPU bit for bit
}

// ndc_rays (nerf_helpers.py:280-307), op for op: python-float scalars enter every tensor op rounded to fp32
__device__ __forceinline__ void nm_ndc_ray(float near32, float c_w, float c_h, float two_near, float m_two_near,
                                           float (&o)[3], float (&d)[3]) {
    const float t = -(near32 + o[2]) / d[2];
    float q[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) { const float td = t * d[a]; q[a] = o[a] + td; }
    const float o0 = (c_w * q[0]) / q[2];
    const float o1 = (c_h * q[1]) / q[2];
    const float rq = 1.0f / q[2];            // python-float / tensor is Tensor.__rtruediv__ = reciprocal() * scalar
    const float o2 = 1.0f + rq * two_near;
    const float d0 = c_w * (d[0] / d[2] - q[0] / q[2]);
    const float d1 = c_h * (d[1] / d[2] - q[1] / q[2]);
    const float d2 = rq * m_two_near;
    o[0] = o0; o[1] = o1; o[2] = o2;
    d[0] = d0; d[1] = d1; d[2] = d2;
}

__device__ __forceinline__ void nm_gen_ray(const RayGen& g, int64_t ray, float (&o)[3], float (&d)[3]) {
    const int64_t pix = g.first + ray;
    const int row = (int)(pix / g.width), colx = (int)(pix - (int64_t)row * g.width);
    const float x = ((float)colx - (float)(g.width * 0.5)) / g.focal;
    const float y = -((float)row - (float)(g.height * 0.5)) / g.focal;
    const float z = -1.0f;
    const float n = nm_norm3(x, y, z);
    const float dx = x / n, dy = y / n, dz = z / n;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        d[a] = (dx * g.rot[3 * a] + dy * g.rot[3 * a + 1]) + dz * g.rot[3 * a + 2];
        o[a] = g.origin[a];
    }
    if (g.ndc) nm_ndc_ray(g.ndc_near, g.c_w, g.c_h, g.two_near, g.m_two_near, o, d);
}

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
    float* tape_v;           // (n, H/2), rows tape_v_ld floats apart
    uint64_t* mask_h;        // (L, tiles, 64): layers_xyz[0..L-2] then fc_feat; per lane, bit 4*tile+reg = activation > 0
    uint64_t* mask_v;        // (tiles, 64)
    int64_t tiles;           // ceil(n / 16)
    float* tape_encx;        // (n, 64) or null: the positional-encoding row of every sample point, reference column order
    float* tape_encd;        // (n, 64) or null: likewise for the view direction (tuned family only: nm_mlp_tape.d_enc_*)
    int32_t tape_v_ld;       // floats per row of tape_v (nm_mlp_tape.v_stride; H/2 when the caller left it 0)
    int32_t tape_skip_h0;    // tuned taping kernels: do not write tape_h[0] (nm_mlp_tape.skip_h0: layer1's output, which a backward that
                             // takes layer1's and layers_xyz[0]'s gradients by linearity never reads)
    RayGen gen;              // VIEW: rays generated from the pose (c = t as in RAYS; a, b unused)
    // generic-shape kernels only (mlp_device_g.h): the encodings' run-time description
    const void* g_tab;       // device: GEncArg[2][96] (xyz, dir; two parts of 48): coordinate and frequency band of every encoding argument
    int32_t g_nsx, g_idx, g_chx;   // xyz: k-steps that carry arguments, 1 = an identity step follows, whole chunks in the stream
    int32_t g_nsd, g_idd, g_chd;   // direction encoding likewise (g_chd = 0: no encoded direction columns at all)
    int32_t g_h, g_hd;             // the REAL widths hidden_size and hidden_size // 2: row strides of the generic training tape
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
    // generic-shape backward (mlp_device_g.h): ReLU' comes from the taped activations instead of masks
    const float* tape_h;     // (L, n, H)
    const float* tape_feat;  // (n, H)
    const float* tape_v;     // (n, H / 2)
    int32_t g_h, g_hd;       // real widths (row strides)
    int32_t stop_at_xyz0;    // tuned delta kernel: do not apply layers_xyz[0]^T (d_h[0] is not produced; NM_BACKWARD_STOP_AT_XYZ0)
};

// Flat addressing of the trainable tensors (index maps of the packed blob: tensor id << 24 | element)
enum TensorId : int {
    T_L1W = 0, T_L1B = 1, T_XYZ0 = 2 /* + 2i weight, + 2i + 1 bias */, T_FEATW = 66, T_FEATB, T_ALPHAW, T_ALPHAB,
    T_DIRW, T_DIRB, T_RGBW, T_RGBB, T_COUNT
};
struct WeightPtrs { const float* p[T_COUNT]; };

// host-side description of one template instantiation of the fused forward kernel
struct MlpPlan {
    int H, FX, FD, NW, KCH, variant;
    int ring_bytes;
    bool lds_bias;
    void (*kernel)(const MlpArgs, const int, const int);
    int wg_samples;      // samples one workgroup evaluates per iteration
    int wg_per_cu;       // workgroups co-resident on a CU
    void (*kernel_flat)(const MlpArgs, const int, const int);   // the FLAT instantiation (use_viewdirs = 0 networks), or null
    int generic_nt;      // 0: a tuned plan for exactly (H, FX, FD) | NT: the generic family's width class (mlp_device_g.h), H = 16 NT
    void (*kernel_tape)(const MlpArgs, const int, const int);        // generic family: the taping forward ...
    void (*kernel_bwd)(const MlpBwdArgs, const int, const int);      // ... and the delta kernel (null for tuned plans: nerf_train.hip)
};

// hipFuncAttributeMaxDynamicSharedMemorySize of a kernel, raised once per (device, kernel) and whenever a launch needs more --
// not on every launch --, under a lock (the attribute is per device; two host threads would otherwise race on it).
int ensure_dynamic_lds(const void* kernel, int bytes);

// The general weight-gradient kernel (nerf_dw_g.hip) as a plain GEMM C = A^T B with a short contraction and a wide output:
// the engine of the layer-wise network path (nerf_layerwise.hip).  `partial` receives the (out_pad x in_pad) result.
struct DwgGemmGeometry { int wa, wb, ta, tb, nba, nbb, out_pad; int64_t in_pad; };
DwgGemmGeometry dwg_gemm_geometry(int out, int64_t in);
struct DwgEpilogue { float* out; const float* bias; const float* mask; int64_t ld; int act; };   // act: 0 none, 1 ReLU, 2 sigmoid
int dwg_gemm(const float* A, int out, int lda, const float* B, int64_t in, int64_t ldb, int rows, float* partial, hipStream_t stream,
             const DwgEpilogue* epilogue = nullptr);

// fused MLP over rays generated from a camera pose (mlp_api.hip; used by the render path in ray_ops.hip)
int nm_mlp_eval_view_internal(nm_mlp* m, const RayGen* gen, const float* d_t, int64_t rays, int32_t samples,
                              float* d_radiance, hipStream_t stream);

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
    // opt-in "bf16x3" precision (mlp_device_b3.h): the same parameters as three bf16 planes per (k-block, tile) unit
    int precision;           // NM_PREC_F32 | NM_PREC_BF16X3
    void* d_stream_b3;       // units x 3 KiB
    float* d_tmp_b3;         // gathered fp32 image [unit][lane][8] the planes are split from
    int32_t* d_index_b3;
    size_t b3_units;
    void* d_enc_tab;         // generic plans: GEncArg[2][48]
    void* lw;                // layer-wise path (nerf_layerwise.h: LwNet*): networks beyond the fused families' limits, else null
    // stale-parameter guard (nm_mlp_weights_current): [0] = checksum of the packed image the LAST gather wrote, [1] = scratch
    // of a verification pass over the caller's live tensors (device, 2 x 8 bytes); how many gathers ran since create
    unsigned long long* d_check;
    int64_t refresh_count;
    // float offset, in the packed image, of the plain copies nm_mlp_linear_layer1_finish reads: layers_xyz[0].weight (H, H),
    // layer1.weight^T (dx, H), layer1.bias (H); 0: none (layer-wise path, one-layer networks)
    size_t plain_off;
};
