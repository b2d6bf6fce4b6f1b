// Fused FlexibleNeRFModel forward for gfx950 (MI355X): positional encoding -> trunk -> sigma /
// feature -> view branch -> rgb in ONE kernel, fp32 MFMA (v_mfma_f32_16x16x4_f32), activations
// never leave registers, weights streamed L2 -> LDS with global_load_lds (direct-to-LDS DMA).
//
// The kernel template itself lives in mlp_device.h (shared with the training kernels of nerf_train.hip);
// this file holds the plan table and the launcher.
// Replaces (reference file:line under /root/reference/src):
//   nerf/modules.py:26-34      PositionalEncoding.forward      (expand, mul, view, sin, cos, cat)
//   nerf/models.py:60-80       FlexibleNeRFModel.forward       (12x addmm, 9x relu, 3x cat, sigmoid)
//   models/model_helpers.py:32-35 intervals_to_ray_points      (RAYS mode prologue)
//   mesh_nerf.py:37-40         grid point generation           (GRID mode prologue)
//
// Dataflow.  The network is evaluated transposed: out^T[feature][sample] = W[feature][k] * act^T[k][sample].
// One wave owns 16 samples (the 16 MFMA columns).  For v_mfma_f32_16x16x4_f32
//   A (weights): lane l holds W[row = l&15][k = l>>4]           (one VGPR)
//   B (activ.) : lane l holds act[k = l>>4][sample = l&15]      (one VGPR)
//   D          : lane l, reg r holds out[row = 4*(l>>4)+r][sample = l&15]
// so D of tile nt, register r is exactly the B operand of the NEXT layer's k-step s = 4*nt + r, in
// which lane group g = l>>4 supplies input feature k = 16*nt + 4*g + r.  The packer (mlp_api.hip: an
// index map built once on the host + one gather kernel) permutes the weight columns accordingly, which
// is why bias+ReLU'd accumulators feed the next layer's MFMAs directly: no LDS round trip, no transposes, no HBM traffic for
// activations.  Weights (2.4 MB / model, L2-resident) are the only streamed operand:
// every workgroup pulls the same linear "A-operand stream" through a 2-deep LDS ring in 8-k-step
// chunks; each lane reads its operands for 4 consecutive output tiles with one conflict-free
// ds_read_b128.
//
// Roofline: MFMA (fp32 matrix peak 157.3 TFLOP/s).  593 408 MAC per sample for the 8x256 net; the
// kernel issues 9 280 MFMAs (1 024 MAC each) per 16-sample tile = 99.9 % useful work.
#include <stdlib.h>

#include <iterator>
#include <mutex>
#include <vector>

#include "nm_internal.h"
#include "mlp_device.h"
#include "mlp_device_r3.h"
#include "mlp_device_b3.h"
#include "mlp_device_g.h"
#ifdef NM_ABLATIONS
#include "mlp_device_g2.h"
#include "mlp_device_b3w.h"
#endif

namespace nm {


// ---- host side: plan table + launcher ------------------------------------------------------
template <int H, int FX, int FD, int NW, int KCH, bool PIPE, bool KEEP_ENC, bool LBIAS, bool SPREAD = false, int ABL = 0>
static MlpPlan make_plan(int variant) {
    return MlpPlan{H, FX, FD, NW, KCH, variant, 2 * Net<H, FX, FD, KCH>::LDSBUF, LBIAS,
                   &mlp_kernel<H, FX, FD, NW, KCH, PIPE, KEEP_ENC, LBIAS, SPREAD, ABL, false>, NW * 16, 8 / NW,
                   (ABL == 0 && LBIAS) ? &mlp_kernel<H, FX, FD, NW, KCH, PIPE, KEEP_ENC, LBIAS, SPREAD, ABL, false, true> : nullptr, 0,
                   nullptr, nullptr};
}

template <int H, int FX, int FD, int NW, int KCH, int STAG, int ABL = 0>
static MlpPlan make_plan3(int variant) {
    return MlpPlan{H, FX, FD, NW, KCH, variant, 3 * Net<H, FX, FD, KCH>::LDSBUF, true, &mlp_kernel3<H, FX, FD, NW, KCH, STAG, ABL>,
                   NW * 16, 8 / NW, ABL == 0 ? &mlp_kernel3<H, FX, FD, NW, KCH, STAG, ABL, true> : nullptr, 0, nullptr, nullptr};
}

// variant 0 is the production choice and the ONLY one in libnerfmeshes_hip.so.  The others exist for within-process
// A/B runs (scripts/bench_mlp.py) and are compiled only with -DNM_ABLATIONS into a separate library
// (libnerfmeshes_hip_ablations.so, `python -m nerfmeshes_amd.build --ablations`), where NM_MLP_VARIANT=<n> selects
// them; the product library never reads that variable, so an inherited environment cannot change its results.
static const MlpPlan g_tuned_plans[] = {
    // production: the round-2 kernel (3-slot ring, operand stream across boundaries, staggered scalar-addressed DMA)
    // for the 256- and 128-wide networks; 64-wide networks (2-tile view layer) keep the round-1 dataflow
    make_plan3<256, 10, 4, 8, 8, 1>(0),
    make_plan3<128, 10, 4, 8, 8, 1>(0),
    make_plan<64, 10, 4, 8, 8, true, true, true>(0),
    make_plan3<256, 6, 4, 8, 8, 1>(0),
    make_plan3<128, 6, 4, 8, 8, 1>(0),
    make_plan<64, 6, 4, 8, 8, true, true, true>(0),
#ifdef NM_ABLATIONS
    // round 1 (profiles/r01_mlp_variants.json; all with the round-1 DMA form, ABL bit 8): v10 = round-1 production 141.1
    make_plan<256, 10, 4, 8, 8, true, true, true, false, 8>(10),
    make_plan<256, 10, 4, 8, 16, true, true, true, false, 8>(1),     // 16-k-step chunks: 137.6
    make_plan<256, 10, 4, 8, 8, false, true, false, false, 8>(2),    // round-1 first version (no prefetch, L2 biases): 133.7
    make_plan<256, 10, 4, 8, 8, true, false, true, false, 8>(3),     // encodings recomputed at the skip layer: ~135
    make_plan<256, 10, 4, 8, 8, true, true, false, false, 8>(4),     // prefetch only, biases from L2: 138.3
    make_plan<256, 10, 4, 4, 8, true, true, true, false, 8>(5),      // 4-wave workgroups, two per CU (decoupled barriers): 132.3
    make_plan<256, 10, 4, 4, 8, true, true, true>(6),                // ... with the scalar-addressed DMA: 143.4 (= variant 9: decoupling buys nothing)
    // timing-only ablations (WRONG results): 1 = no sincos, 2 = no barrier, 4 = no weight DMA
    make_plan<256, 10, 4, 8, 8, true, true, true, false, 1>(11),
    make_plan<256, 10, 4, 8, 8, true, true, true, false, 2>(12),
    make_plan<256, 10, 4, 8, 8, true, true, true, false, 4>(14),
    make_plan<256, 10, 4, 8, 8, true, true, true, false, 6>(16),
    make_plan<256, 10, 4, 8, 8, true, true, true, false, 7>(17),
    // round 2 (profiles/r02_mlp_variants.json): 3-slot ring, DMA two chunks ahead, operand stream across boundaries
    make_plan<256, 10, 4, 8, 8, true, true, true>(9),  // round-1 dataflow (2-slot ring) with the scalar-addressed DMA
    make_plan3<256, 10, 4, 8, 8, 0>(21),        // all waves at block 0
    make_plan3<256, 10, 4, 8, 8, 1, 8>(31),     // v20 with the round-1 DMA form (address VGPRs)
    make_plan3<256, 10, 4, 8, 8, 1, 8 | 1>(22), // timing only (WRONG results): ... barriers do not wait for the DMA
    make_plan3<256, 10, 4, 8, 8, 1, 2>(23),     // timing only (WRONG results): no weight DMA at all
    make_plan3<256, 10, 4, 8, 8, 1, 4>(42),     // timing only (WRONG results): round-1 DMA form issued for one lane
#endif
};

// the generic-shape family (mlp_device_g.h), instantiated in nerf_mlp_generic_{a..e}.hip, in ascending width
void generic_plans_a(std::vector<MlpPlan>&);
void generic_plans_b(std::vector<MlpPlan>&);
void generic_plans_c(std::vector<MlpPlan>&);
void generic_plans_d(std::vector<MlpPlan>&);
void generic_plans_e(std::vector<MlpPlan>&);
void generic_plans_s(std::vector<MlpPlan>&);
void generic_plans_a_long(std::vector<MlpPlan>&);      // the same classes with two-part encoding stages (variant G_LONG_VARIANT)
void generic_plans_b_long(std::vector<MlpPlan>&);
void generic_plans_c_long(std::vector<MlpPlan>&);
void generic_plans_d_long(std::vector<MlpPlan>&);
void generic_plans_s_long(std::vector<MlpPlan>&);
void generic_plans_s_long_upper(std::vector<MlpPlan>&);

static const std::vector<MlpPlan>& all_plans() {
    static const std::vector<MlpPlan> plans = [] {
        std::vector<MlpPlan> v(std::begin(g_tuned_plans), std::end(g_tuned_plans));
        generic_plans_a(v); generic_plans_b(v); generic_plans_c(v); generic_plans_d(v); generic_plans_s(v); generic_plans_e(v);
        generic_plans_a_long(v); generic_plans_b_long(v); generic_plans_c_long(v); generic_plans_d_long(v); generic_plans_s_long(v); generic_plans_s_long_upper(v);
#ifdef NM_ABLATIONS
        // experiment (NM_MLP_VARIANT=200 + NM_KERNEL_GENERIC): two 16-sample column tiles per wave (mlp_device_g2.h)
        // (round 5: also the classes of 2 and 3 tiles -- VERDICT r4 item 7 -- compiled for four waves per SIMD (200) and two (201))
        v.push_back(MlpPlan{32, -1, -1, 8, 8, 200, 2 * 8 * 1024, true, &mlp_kernel_g2<2, 8, 8, 4>, 8 * 32, 1, nullptr, 2, nullptr, nullptr});
        v.push_back(MlpPlan{48, -1, -1, 8, 8, 200, 2 * 8 * 1024, true, &mlp_kernel_g2<3, 8, 8, 4>, 8 * 32, 1, nullptr, 3, nullptr, nullptr});
        v.push_back(MlpPlan{32, -1, -1, 8, 8, 201, 2 * 8 * 1024, true, &mlp_kernel_g2<2, 8, 8, 2>, 8 * 32, 1, nullptr, 2, nullptr, nullptr});
        v.push_back(MlpPlan{48, -1, -1, 8, 8, 201, 2 * 8 * 1024, true, &mlp_kernel_g2<3, 8, 8, 2>, 8 * 32, 1, nullptr, 3, nullptr, nullptr});
        v.push_back(MlpPlan{64, -1, -1, 8, 8, 200, 2 * 8 * 1024, true, &mlp_kernel_g2<4, 8, 8>, 8 * 32, 1, nullptr, 4, nullptr, nullptr});
        v.push_back(MlpPlan{128, -1, -1, 8, 8, 200, 2 * 8 * 2 * 1024, true, &mlp_kernel_g2<8, 8, 8>, 8 * 32, 1, nullptr, 8, nullptr, nullptr});
#endif
        return v;
    }();
    return plans;
}

// opt-in bf16x3 precision (mlp_device_b3.h)
struct B3Plan {
    int H, FX, FD;
    void (*kernel)(const MlpArgs, const int, const int);
    void (*kernel_w)(const MlpArgs, const int, const int);      // ablation library: two column tiles per wave (mlp_device_b3w.h)
    int chunk_units;                                            // units per ring slot (a stage must span two chunks)
};
static const B3Plan g_b3_plans[] = {
#ifdef NM_ABLATIONS
    {256, 10, 4, &mlp_kernel_b3<256, 10, 4, 8>, &mlp_kernel_b3w<256, 10, 4, 4>, B3_CHUNK_UNITS},
    {256, 6, 4, &mlp_kernel_b3<256, 6, 4, 8>, &mlp_kernel_b3w<256, 6, 4, 4>, B3_CHUNK_UNITS},
#else
    {256, 10, 4, &mlp_kernel_b3<256, 10, 4, 8>, nullptr, B3_CHUNK_UNITS},
    {256, 6, 4, &mlp_kernel_b3<256, 6, 4, 8>, nullptr, B3_CHUNK_UNITS},
#endif
    // round 5: the narrower shipped shapes (the fern configs' 8x128, config 1's 4x64): smaller chunks so that every stage still
    // spans two of them; the same kernel otherwise
    {128, 10, 4, &mlp_kernel_b3<128, 10, 4, 8, 8>, nullptr, 8},
    {128, 6, 4, &mlp_kernel_b3<128, 6, 4, 8, 8>, nullptr, 8},
    {64, 10, 4, &mlp_kernel_b3<64, 10, 4, 8, 3>, nullptr, 3},
    {64, 6, 4, &mlp_kernel_b3<64, 6, 4, 8, 3>, nullptr, 3},
};
static const B3Plan* find_b3_plan(int H, int FX, int FD) {
    for (const B3Plan& p : g_b3_plans)
        if (p.H == H && p.FX == FX && p.FD == FD) return &p;
    return nullptr;
}
bool has_b3_kernel(int H, int FX, int FD) { return find_b3_plan(H, FX, FD) != nullptr; }

// A tuned plan when one is instantiated for exactly this shape (and the input itself is part of both encodings, which the
// tuned kernels' identity k-step assumes is laid out -- its weights may still be zero); otherwise the narrowest class of
// the generic family that holds the network; null only beyond the family's limits (mlp_api.hip says which).
const MlpPlan* find_mlp_plan(int H, int FX, int FD) {
    int want = 0;
#ifdef NM_ABLATIONS
    if (const char* v = getenv("NM_MLP_VARIANT")) want = atoi(v);
#endif
    const MlpPlan* fallback = nullptr;
    for (const MlpPlan& p : all_plans())
        if (!p.generic_nt && p.H == H && p.FX == FX && p.FD == FD) {
            if (p.variant == want) return &p;
            if (p.variant == 0) fallback = &p;
        }
    return fallback;
}

// the narrowest class of the generic family that holds hidden_size H and whose LDS image (weight ring + every bias of an L-layer
// network + heads + tables) fits a CU; `long_encoding`: an encoding of more than G_ENC_STEPS k-steps (16 -- 31 functions) -- the
// instantiations with two-part encoding stages.  Null if there is none (the caller then takes the layer-wise path).
const MlpPlan* find_generic_plan(int H, int L, bool long_encoding) {
    int want = long_encoding ? G_LONG_VARIANT : 0;
#ifdef NM_ABLATIONS
    if (const char* v = getenv("NM_MLP_VARIANT"))
        if (!long_encoding) want = (atoi(v) == 200 || atoi(v) == 201 || atoi(v) == 310) ? atoi(v) : 0;
#endif
    for (const MlpPlan& p : all_plans())
        if (p.generic_nt && p.variant == want && p.H >= H && g_lds_bytes(p.ring_bytes, p.generic_nt, L, p.variant == G_LONG_VARIANT ? G_ENC_PARTS : 1) <= 160 * 1024) return &p;
    return nullptr;
}

int mlp_plan_info(const MlpPlan* p, int* nw) {
    *nw = p->NW;
    return p->generic_nt ? 1000 + p->generic_nt : p->variant;     // generic family: 1000 + width class
}

// The handle's device must be current for the launch (the stream belongs to it): a model on cuda:1 used from a process
// whose current device is cuda:0 is switched to for the call and switched back.
struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int want) {
        int cur = -1;
        if (hipGetDevice(&cur) == hipSuccess && cur != want && hipSetDevice(want) == hipSuccess) prev = cur;
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

int launch_mlp(const nm_mlp* m, const MlpArgs& args, int density_only, hipStream_t stream) {
    const MlpPlan* p = m->plan;
    if (args.n <= 0) return 0;
    DeviceGuard guard(m->device);
    const int L = m->desc.num_layers, H = m->desc.hidden_size;
    NM_REQUIRE(density_only != 2 || (m->precision == NM_PREC_F32 && p->kernel_flat), "no kernel for use_viewdirs=0 networks in this plan");
    auto kernel = density_only == 2 ? p->kernel_flat : p->kernel;
    if (m->precision == NM_PREC_BF16X3) {
        const B3Plan* b = find_b3_plan(H, m->desc.num_encoding_fn_xyz, m->desc.num_encoding_fn_dir);
        NM_REQUIRE(b && m->d_stream_b3, "no bf16x3 kernel for this network");
        const int lds_bytes = 3 * b->chunk_units * B3_UNIT + (((H * (1 + L) + H / 2 + 4 + H + 3 * H / 2 + 32) * 4 + 255) & ~255);   // + the two band tables
        NM_REQUIRE(lds_bytes <= 160 * 1024, "LDS budget exceeded (bf16x3 ring + bias cache): too many layers");
        auto b3_kernel = b->kernel;
        unsigned b3_threads = 512;
#ifdef NM_ABLATIONS
        if (const char* v = getenv("NM_MLP_VARIANT"))       // experiment: 4 waves x 32 samples, the same 128-sample workgroup
            if (atoi(v) == 300) { b3_kernel = b->kernel_w; b3_threads = 256; }
#endif
        NM_HIP_CHECK(hipFuncSetAttribute((const void*)b3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
        MlpArgs a = args;
        a.wstream = static_cast<const char*>(m->d_stream_b3);
        const int64_t wg_iters = (a.n + 127) / 128;
        int64_t grid = wg_iters < (int64_t)m->num_cus * 4 ? wg_iters : (int64_t)m->num_cus * 4;
        if (wg_iters > grid) {
            const int64_t rounds = (wg_iters + grid - 1) / grid;
            grid = (wg_iters + rounds - 1) / rounds;
        }
        hipLaunchKernelGGL(b3_kernel, dim3((unsigned)grid), dim3(b3_threads), lds_bytes, stream, a, L, density_only);
        NM_HIP_CHECK(hipGetLastError());
        return 0;
    }
    // bias cache + fc_alpha + fc_rgb rows; a use_viewdirs = 0 network (mode 2) keeps three H-wide fc_out rows where fc_rgb's
    // three H/2-wide ones go
    const int rgb_floats = density_only == 2 ? 3 * H : 3 * H / 2;
    int lds_bytes = p->ring_bytes + (p->lds_bias ? (((H * (1 + L) + H / 2 + 4 + H + rgb_floats) * 4 + 255) & ~255) : 0);
    if (p->generic_nt) {    // padded widths, both head layouts, the two argument tables (mlp_device_g.h)
        lds_bytes = g_lds_bytes(p->ring_bytes, p->generic_nt, L, p->variant == G_LONG_VARIANT ? G_ENC_PARTS : 1);
    }
    NM_REQUIRE(lds_bytes <= 160 * 1024, "LDS budget exceeded (ring + bias cache)");
    // the dynamic-LDS attribute is per device: tracked per (device, plan)
    // (two host threads creating / launching models race on it otherwise: the table is read and written under a lock)
    static std::vector<int> attr_bytes(64 * 2 * all_plans().size(), 0);
    static std::mutex attr_lock;
    const int idx = 2 * (int)(p - all_plans().data()) + (density_only == 2 ? 1 : 0), dev = (m->device >= 0 && m->device < 64) ? m->device : 0;
    {
        std::lock_guard<std::mutex> hold(attr_lock);
        int& have = attr_bytes[(size_t)dev * 2 * all_plans().size() + idx];
        if (have < lds_bytes) {
            NM_HIP_CHECK(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
            have = lds_bytes;
        }
    }
    const int64_t wg_iters = (args.n + p->wg_samples - 1) / p->wg_samples;
    const int64_t resident = (int64_t)m->num_cus * p->wg_per_cu;   // workgroups co-resident on the chip
    // persistent-style launch: a few workgroups per CU queue so the tail is balanced.
    int64_t grid = wg_iters < resident * 4 ? wg_iters : resident * 4;
    // keep the per-workgroup iteration count even across the grid where possible
    if (wg_iters > grid) {
        const int64_t rounds = (wg_iters + grid - 1) / grid;
        grid = (wg_iters + rounds - 1) / rounds;
    }
    hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(p->NW * 64), lds_bytes, stream, args,
                       (int)m->desc.num_layers, density_only);
    NM_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace nm
