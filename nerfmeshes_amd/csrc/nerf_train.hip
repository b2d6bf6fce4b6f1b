// Training path of the fused MLP for gfx950: what autograd does for
//   /root/reference/src/models/model_nerf.py:88-151   NeRFModel.training_step (loss.backward() through
//   /root/reference/src/nerf/models.py:60-80          FlexibleNeRFModel.forward)
// split the MI355X way:
//   * forward  = the SAME fused kernel as inference with TAPE=true: every post-activation leaves the registers once,
//     as fp32 rows (operands of the weight-gradient GEMMs) plus one 64-bit ReLU mask per lane and layer.
//   * backward = mlp_backward_kernel below: the delta of every layer, propagated through the TRANSPOSED weights with
//     the same register-resident MFMA chain (D layout of one stage == B layout of the next; the transposed
//     A-operand stream is a second index map over the same parameters, mlp_api.hip).  It reads 16 B + 8 B x layers
//     per sample and writes the deltas once.
//   * weight gradients dW = delta^T @ activation are plain (H x n) @ (n x H) GEMMs over those rows: library work
//     (rocBLAS through torch.mm in hip_ops.py), per the "library GEMMs only for plain GEMMs" rule.
// Roofline: MFMA.  557 056 MAC / sample for the 8x256 net (vs 593 408 forward).
#include <cstdlib>
#include <map>
#include <mutex>
#include <utility>

#include "nm_internal.h"
#include "mlp_device.h"
#ifdef NM_ABLATIONS
#include <cstdlib>
#include "mlp_device_r3.h"
#endif
#include "mlp_device_g.h"
#include "nerf_layerwise.h"

namespace nm {

template <int NT>
__device__ __forceinline__ void masked_operand(const f32x4 (&acc)[NT], uint64_t mask, float (&op)[4 * NT]) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) op[4 * nt + r] = ((mask >> (4 * nt + r)) & 1u) ? acc[nt][r] : 0.0f;
}

// FLAT: a network without view directions (models.py:77-79: the trunk ends in fc_out, 4 rows): no layers_dir / fc_feat stages;
// the delta at the trunk's output is fc_out^T applied to the four head deltas on the VALU.
template <int H, int NW, int KCH, bool FLAT = false>
__global__ __launch_bounds__(NW * 64, H <= 128 ? 4 : 2) void mlp_backward_kernel(const MlpBwdArgs args, const int num_layers) {
    using N = Net<H, 10, 4, KCH>;   // only the hidden-width constants are used here
    extern __shared__ __attribute__((aligned(16))) char lds[];
    float* lds_walpha = reinterpret_cast<float*>(lds + 2 * N::LDSBUF);   // [4][H/4]
    float* lds_wrgb = lds_walpha + H;                                    // [3][4][H/8]
    for (int i = threadIdx.x; i < H; i += NW * 64) lds_walpha[i] = args.walpha[i];
    for (int i = threadIdx.x; i < (FLAT ? 3 * H : 3 * H / 2); i += NW * 64) lds_wrgb[i] = args.wrgb[i];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, col = lane & 15;
    const int L = num_layers;
    // first chunk of the first stage: layers_dir.0^T, or (FLAT) layers_xyz[L-2]^T
    constexpr int FIRST = FLAT ? N::LDSBUF : (N::KD < KCH ? N::KD : KCH) * N::STEP;

    const int64_t wg_iters = (args.n + NW * 16 - 1) / (NW * 16);
    int par = 0;
    const char* gw = args.wstream;
    if ((int64_t)blockIdx.x < wg_iters) stream_to_lds<NW>(gw, lds, FIRST, wave, lane);

    for (int64_t it = blockIdx.x; it < wg_iters; it += gridDim.x) {
        const bool has_next = it + gridDim.x < wg_iters;
        const int64_t sample = (it * NW + wave) * 16 + col;
        const bool valid = sample < args.n;
        const int64_t sidx = valid ? sample : args.n - 1;
        const int64_t tile = it * NW + wave;
        const bool tile_ok = tile < args.tiles;
        const uint64_t* mrow = args.mask_h + tile * 64 + lane;      // + layer * tiles * 64
        const int64_t mstride = args.tiles * 64;
        const float dummy[1] = {0.0f};
        __syncthreads();   // LDS caches filled; first chunk resident (its DMA was issued one tile earlier)

        // ---- head: sigmoid', fc_rgb^T on the VALU (models.py:75)
        const f32x4 go = *reinterpret_cast<const f32x4*>(args.grad_out + 4 * sidx);
        const f32x4 y = *reinterpret_cast<const f32x4*>(args.radiance + 4 * sidx);
        float drgb[3];
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) drgb[ch] = go[ch] * (y[ch] * (1.0f - y[ch]));
        const float dsigma = go[3];
        if (valid && g == 0) {
            const f32x4 o4 = {drgb[0], drgb[1], drgb[2], dsigma};
            *reinterpret_cast<f32x4*>(args.d_last + 4 * sample) = o4;
        }
        // every delta is written while the NEXT stage consumes it (gemm_stage STORE; measured: the 16-instruction
        // bursts cost 10 % -- 3.49 ms with them, 3.15 ms with the stores ablated)
        float* const d_row = valid ? args.d_h + sample * H + 4 * g : nullptr;

        f32x4 acc[N::NT];
        float in[N::KH];
        if constexpr (FLAT) {
            // ---- fc_out^T: delta at the output of layers_xyz[L-2] = sum over the four rows of fc_out (fc_alpha's operand layout)
            const float* wa = lds_walpha + g * (H / 4);
#pragma unroll
            for (int nt = 0; nt < N::NT; ++nt) {
                f32x4 w4 = *reinterpret_cast<const f32x4*>(wa + 4 * nt);
                f32x4 a = {w4[0] * dsigma, w4[1] * dsigma, w4[2] * dsigma, w4[3] * dsigma};
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) {
                    w4 = *reinterpret_cast<const f32x4*>(lds_wrgb + ch * H + g * (H / 4) + 4 * nt);
#pragma unroll
                    for (int r = 0; r < 4; ++r) a[r] = fmaf(w4[r], drgb[ch], a[r]);
                }
                acc[nt] = a;
            }
            const uint64_t m = tile_ok ? mrow[(int64_t)(L - 2) * mstride] : 0;
            masked_operand<N::NT>(acc, m, in);
        } else {
        const uint64_t mv = tile_ok ? args.mask_v[tile * 64 + lane] : 0;
        float dv[N::KD];
#pragma unroll
        for (int s = 0; s < N::KD; ++s) {
            float a = lds_wrgb[(0 * 4 + g) * N::KD + s] * drgb[0];
            a = fmaf(lds_wrgb[(1 * 4 + g) * N::KD + s], drgb[1], a);
            a = fmaf(lds_wrgb[(2 * 4 + g) * N::KD + s], drgb[2], a);
            dv[s] = ((mv >> s) & 1u) ? a : 0.0f;
        }
        // ---- layers_dir.0^T (hidden columns): delta at relu(fc_feat) -> masked -> delta at fc_feat's output
        {
#pragma unroll
            for (int nt = 0; nt < N::NT; ++nt) acc[nt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            const uint64_t m = tile_ok ? mrow[(int64_t)(L - 1) * mstride] : 0;
            gemm_stage<N::NT, N::KD, 0, NW, N::LDSBUF, KCH, true, false, 0, true>(
                acc, dv, dummy, gw, gw + N::KD * N::STEP, N::LDSBUF, lds, par, wave, lane,
                valid ? args.d_v + sample * (H / 2) + 4 * g : nullptr);
            gw += N::KD * N::STEP;
            masked_operand<N::NT>(acc, m, in);
        }
        // ---- fc_feat^T + fc_alpha^T: delta at the output of layers_xyz[L-2]
        {
            const float* wa = lds_walpha + g * (H / 4);
#pragma unroll
            for (int nt = 0; nt < N::NT; ++nt) {
                const f32x4 w4 = *reinterpret_cast<const f32x4*>(wa + 4 * nt);
                acc[nt] = f32x4{w4[0] * dsigma, w4[1] * dsigma, w4[2] * dsigma, w4[3] * dsigma};
            }
            const uint64_t m = tile_ok ? mrow[(int64_t)(L - 2) * mstride] : 0;
            gemm_stage<N::NT, N::KH, 0, NW, N::LDSBUF, KCH, true, false, 0, true>(
                acc, in, dummy, gw, gw + N::KH * N::STEP, N::LDSBUF, lds, par, wave, lane,
                valid ? args.d_feat + sample * H + 4 * g : nullptr);
            gw += N::KH * N::STEP;
            masked_operand<N::NT>(acc, m, in);
        }
        }
        // ---- layers_xyz[i]^T, i = L-2 .. 0: delta at the input of layers_xyz[i] (x[i-1] post-ReLU, or layer1's output).
        //      stop_at_xyz0: layers_xyz[0]^T is left out -- layer1 has no activation (models.py:62), so the caller gets its gradient
        //      from the delta at layers_xyz[0] by linearity (train_ops.backward: W0^T applied once to delta^T @ encoding) -- and the
        //      chain ends with d_h[1]
        const int last_i = args.stop_at_xyz0 ? 1 : 0;
#pragma unroll 1
        for (int i = L - 2; i >= last_i; --i) {
#pragma unroll
            for (int nt = 0; nt < N::NT; ++nt) acc[nt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            uint64_t m = ~uint64_t(0);
            if (i > 0) m = tile_ok ? mrow[(int64_t)(i - 1) * mstride] : 0;
            const char* tsrc = gw + N::KH * N::STEP;
            int tbytes = N::LDSBUF;
            if (i == last_i) { tsrc = args.wstream; tbytes = has_next ? FIRST : 0; }
            gemm_stage<N::NT, N::KH, 0, NW, N::LDSBUF, KCH, true, false, 0, true>(
                acc, in, dummy, gw, tsrc, tbytes, lds, par, wave, lane,
                d_row ? d_row + (int64_t)(i + 1) * args.n * H : nullptr);   // the delta this stage consumes
            gw += N::KH * N::STEP;
            masked_operand<N::NT>(acc, m, in);
        }
        store_rows<N::NT>(args.d_h + (int64_t)last_i * args.n * H, H, sample, valid, in, g);   // the chain's last delta: nothing follows
        gw = args.wstream;
    }
}

// ---- positional encodings as rows, reference column order (operands of the layer1 / skip / view weight gradients)
struct EncodeArgs {
    const float* origins; const float* dirs; const float* t;
    int64_t n; int32_t samples, origins_per_ray;
    int32_t fx, fd, include_x, include_d;
    float bands_xyz[MAX_FREQ_XYZ];
    float bands_dir[MAX_FREQ_DIR];
    float* enc_x; float* enc_d;
    int32_t stride_x, stride_d;   // floats per output row (>= the encoding width; the tail of a row is left untouched)
};

// PositionalEncoding rows for the weight-gradient kernels: WHOLE rows of the given strides (zero padding included).  A thread
// serves one work item of one sample of one encoding (blockIdx.y: 0 = xyz, 1 = direction): item w < 3 F is ARGUMENT w -- one
// sincosf, both results written (sin at column base + w, cos at base + 3 F + w: neighbouring threads write neighbouring
// floats) --, the items behind are the remaining columns one by one (the input itself, the zero padding).  Index arithmetic in
// 32 bits whenever the launch allows it (a 64-bit division is ~100 instructions; the first version spent more time dividing
// than encoding).  sincosf is evaluated on the same product as encode<>(): the values are bit-identical to the forward kernel's.
__global__ __launch_bounds__(256) void encode_samples_rows_kernel(const EncodeArgs args, const int narrow) {
    const bool dir = blockIdx.y == 1;
    float* out = dir ? args.enc_d : args.enc_x;
    if (!out) return;
    const int F = dir ? args.fd : args.fx, include = dir ? args.include_d : args.include_x, stride = dir ? args.stride_d : args.stride_x;
    const int items = stride - 3 * F;                       // 3 F arguments + (stride - 6 F) plain columns
    const float* bands = dir ? args.bands_dir : args.bands_xyz;
    const int base = include ? 3 : 0;
    const int64_t total = args.n * items;
    // a grid-stride loop over the work items: a few thousand workgroups that stay, not one per 256 items (the dispatcher, not
    // the memory system, bounded the one-shot form: 10^5 workgroups of a few instructions each)
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    int64_t i, ray;
    int w;
    if (narrow) {                                           // n * items < 2^31
        const unsigned iu = (unsigned)e / (unsigned)items;
        w = (int)((unsigned)e - iu * (unsigned)items);
        i = iu;
        ray = iu / (unsigned)args.samples;
    } else {
        i = e / items;
        w = (int)(e - i * items);
        ray = i / args.samples;
    }
    float x[3];
    if (dir) {
#pragma unroll
        for (int k = 0; k < 3; ++k) x[k] = args.dirs[3 * ray + k];
    } else {
        const float t = args.t[i];
        const float* o = args.origins + (args.origins_per_ray ? 3 * ray : 0);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float dt = args.dirs[3 * ray + k] * t;
            x[k] = o[k] + dt;
        }
    }
    float* row = out + i * stride;
    if (w < 3 * F) {
        float sv, cv;
        sincosf(x[w / F] * bands[w % F], &sv, &cv);
        row[base + w] = sv;
        row[base + 3 * F + w] = cv;
    } else {
        const int c = w - 3 * F;                             // the c-th column that is not a sine / cosine
        if (c < base) row[c] = c == 0 ? x[0] : (c == 1 ? x[1] : x[2]);
        else row[6 * F + c] = 0.0f;
    }
    }
}

// ---- plan tables -------------------------------------------------------------------------------------------
struct TrainPlan {
    int H, FX, FD;
    int ring_slots;                                                // LDS ring: 3 (mlp_kernel3 dataflow) or 2
    void (*forward)(const MlpArgs, const int, const int);
    void (*forward_flat)(const MlpArgs, const int, const int);     // the taping kernel that also serves use_viewdirs = 0 networks
};
template <int H, int FX, int FD>
static TrainPlan make_train_plan() {
    return TrainPlan{H, FX, FD, 2, &mlp_kernel<H, FX, FD, 8, KC, true, true, true, false, 0, true>,
                     &mlp_kernel<H, FX, FD, 8, KC, true, true, true, false, 0, true, true>};
}
static const TrainPlan g_train_plans[] = {
    make_train_plan<256, 10, 4>(), make_train_plan<128, 10, 4>(), make_train_plan<64, 10, 4>(),
    make_train_plan<256, 6, 4>(),  make_train_plan<128, 6, 4>(),  make_train_plan<64, 6, 4>(),
};
#ifdef NM_ABLATIONS
// experiment (ablation library, NM_MLP_VARIANT=3): the 256- and 128-wide networks tape on the inference kernel's dataflow
// (3-slot ring, operand stream across chunk and stage boundaries, staggered DMA: mlp_device_r3.h) -- same tape, same bits,
// measured slower (see mlp_kernel3)
template <int H, int FX, int FD>
static TrainPlan make_train_plan3() {
    return TrainPlan{H, FX, FD, 3, &mlp_kernel3<H, FX, FD, 8, KC, 1, 0, false, true>, &mlp_kernel3<H, FX, FD, 8, KC, 1, 0, true, true>};
}
static const TrainPlan g_train_plans3[] = {
    make_train_plan3<256, 10, 4>(), make_train_plan3<128, 10, 4>(), make_train_plan3<256, 6, 4>(), make_train_plan3<128, 6, 4>(),
};
#endif

struct BwdPlan {
    int H;
    void (*backward)(const MlpBwdArgs, const int);
    void (*backward_flat)(const MlpBwdArgs, const int);
};
static const BwdPlan g_bwd_plans[] = {
    {256, &mlp_backward_kernel<256, 8, KC>, &mlp_backward_kernel<256, 8, KC, true>},
    {128, &mlp_backward_kernel<128, 8, KC>, &mlp_backward_kernel<128, 8, KC, true>},
    {64, &mlp_backward_kernel<64, 8, KC>, &mlp_backward_kernel<64, 8, KC, true>},
};

static unsigned persistent_grid(int64_t wg_iters, int num_cus) {
    const int64_t resident = num_cus;   // one 8-wave workgroup per CU (2 waves / SIMD)
    int64_t grid = wg_iters < resident * 4 ? wg_iters : resident * 4;
    if (wg_iters > grid) {
        const int64_t rounds = (wg_iters + grid - 1) / grid;
        grid = (wg_iters + rounds - 1) / rounds;
    }
    return (unsigned)grid;
}

struct DeviceGuardT {        // the handle's device current for the call (as nerf_mlp.hip's DeviceGuard)
    int prev = -1;
    explicit DeviceGuardT(int want) {
        int cur = -1;
        if (hipGetDevice(&cur) == hipSuccess && cur != want && hipSetDevice(want) == hipSuccess) prev = cur;
    }
    ~DeviceGuardT() { if (prev >= 0) (void)hipSetDevice(prev); }
};

static int set_lds(const void* fn, int bytes) { return ensure_dynamic_lds(fn, bytes); }

}  // namespace nm

using namespace nm;

extern "C" {

int nm_mlp_tapes_encodings(const nm_mlp* m) {
    if (!m || m->lw || m->plan->generic_nt != 0) return 0;           // tuned family only (mlp_device.h: store_encoding_row)
    static const bool off = getenv("NM_TAPE_ENCODINGS") && atoi(getenv("NM_TAPE_ENCODINGS")) == 0;   // A/B hook of the tools
    if (off) return 0;
    const nm_mlp_desc& d = m->desc;
    return d.include_input_xyz && (d.use_viewdirs == 0 || d.include_input_dir) && 6 * d.num_encoding_fn_xyz + 3 <= 64 &&
           6 * d.num_encoding_fn_dir + 3 <= 64;
}

int nm_mlp_forward_train(nm_mlp* m, const float* d_origins, int origins_per_ray, const float* d_dirs, const float* d_t,
                         int64_t rays, int32_t samples, const nm_mlp_tape* tape, float* d_radiance, void* stream) {
    NM_REQUIRE(m && d_origins && d_dirs && d_t && tape && d_radiance && rays >= 0 && samples > 0, "bad argument");
    const nm_mlp_desc& d = m->desc;
    const bool flat = d.use_viewdirs == 0;      // models.py:77-79: the tape is the trunk's (d_h, d_mask_h); d_feat / d_v / d_mask_v are not touched
    const bool generic = m->plan->generic_nt != 0 || m->lw;   // generic-shape family / layer-wise path: activation rows only, no ReLU masks
    NM_REQUIRE(tape->d_h && (generic || tape->d_mask_h) && (flat || (tape->d_feat && tape->d_v && (generic || tape->d_mask_v))), "incomplete tape");
    NM_REQUIRE(m->precision == NM_PREC_F32, "training runs in fp32: create the handle with NM_PREC_F32");
    NM_REQUIRE(tape->v_stride >= 0 && (tape->v_stride == 0 || tape->v_stride >= d.hidden_size / 2), "nm_mlp_tape.v_stride is smaller than a row of d_v");
    NM_REQUIRE(!generic || tape->v_stride == 0 || tape->v_stride == d.hidden_size / 2, "only the tuned family writes d_v with a row stride (v_stride)");
    NM_REQUIRE(!tape->skip_h0 || (!generic && d.num_layers >= 2), "nm_mlp_tape.skip_h0 is a tuned-family option");
    if (m->lw) {             // beyond the fused families: layer by layer (nerf_layerwise.hip); the tape is rows, no masks
        MlpArgs a = m->base;
        a.mode = MODE_RAYS;
        a.a = d_origins; a.b = d_dirs; a.c = d_t;
        a.origins_per_ray = origins_per_ray; a.samples = samples;
        a.n = rays * samples; a.out = d_radiance;
        DeviceGuardT guard(m->device);
        return layerwise_forward_train(m, a, tape, static_cast<hipStream_t>(stream));
    }
    if (generic) {
        MlpArgs a = m->base;
        a.mode = MODE_RAYS;
        a.a = d_origins; a.b = d_dirs; a.c = d_t;
        a.origins_per_ray = origins_per_ray; a.samples = samples;
        a.n = rays * samples; a.out = d_radiance;
        if (a.n == 0) return 0;
        a.tape_h = tape->d_h; a.tape_feat = tape->d_feat; a.tape_v = tape->d_v;
        a.tiles = (a.n + 15) / 16;
        const MlpPlan* p = m->plan;
        const int L = d.num_layers;
        const int lds_bytes = g_lds_bytes(p->ring_bytes, p->generic_nt, L, p->variant == G_LONG_VARIANT ? G_ENC_PARTS : 1);
        if (int rc = set_lds((const void*)p->kernel_tape, lds_bytes)) return rc;
        const int64_t wg_iters = (a.n + p->wg_samples - 1) / p->wg_samples;
        hipLaunchKernelGGL(p->kernel_tape, dim3(persistent_grid(wg_iters, m->num_cus)), dim3(p->NW * 64), lds_bytes,
                           static_cast<hipStream_t>(stream), a, L, flat ? 2 : 0);
        NM_HIP_CHECK(hipGetLastError());
        return 0;
    }
    const TrainPlan* plan = nullptr;
    for (const TrainPlan& p : g_train_plans)
        if (p.H == d.hidden_size && p.FX == d.num_encoding_fn_xyz && p.FD == (flat ? 4 : d.num_encoding_fn_dir)) plan = &p;
    NM_REQUIRE(plan, "no training kernel instantiated for this network shape");
#ifdef NM_ABLATIONS
    if (const char* v = getenv("NM_MLP_VARIANT"))
        if (atoi(v) == 3)
            for (const TrainPlan& p : g_train_plans3)
                if (p.H == plan->H && p.FX == plan->FX && p.FD == plan->FD) { plan = &p; break; }
#endif
    MlpArgs a = m->base;
    a.mode = MODE_RAYS;
    a.a = d_origins; a.b = d_dirs; a.c = d_t;
    a.origins_per_ray = origins_per_ray; a.samples = samples;
    a.n = rays * samples; a.out = d_radiance;
    if (a.n == 0) return 0;
    a.tape_h = tape->d_h; a.tape_feat = tape->d_feat; a.tape_v = tape->d_v;
    a.tape_v_ld = tape->v_stride > 0 ? tape->v_stride : d.hidden_size / 2;
    a.tape_skip_h0 = tape->skip_h0 ? 1 : 0;
    a.mask_h = tape->d_mask_h; a.mask_v = tape->d_mask_v;
    a.tiles = (a.n + 15) / 16;
    if (nm_mlp_tapes_encodings(m)) {       // the encoding rows the weight gradients contract with, straight from the registers
        a.tape_encx = tape->d_enc_xyz;
        a.tape_encd = flat ? nullptr : tape->d_enc_dir;
    }
    const int H = d.hidden_size, L = d.num_layers;
    const int ring = plan->ring_slots * KC * (H / 16) * 256;
    const int lds_bytes = ring + (((H * (1 + L) + H / 2 + 4 + H + (flat ? 3 * H : 3 * H / 2)) * 4 + 255) & ~255);
    const auto kernel = flat ? plan->forward_flat : plan->forward;
    if (int rc = set_lds((const void*)kernel, lds_bytes)) return rc;
    const int64_t wg_iters = (a.n + 127) / 128;
    hipLaunchKernelGGL(kernel, dim3(persistent_grid(wg_iters, m->num_cus)), dim3(512), lds_bytes,
                       static_cast<hipStream_t>(stream), a, L, flat ? 2 : 0);
    NM_HIP_CHECK(hipGetLastError());
    return 0;
}

int nm_mlp_backward(nm_mlp* m, int64_t n, const nm_mlp_tape* tape, const float* d_radiance,
                    const float* d_grad_radiance, const nm_mlp_deltas* deltas, void* stream) {
    return nm_mlp_backward_ex(m, n, tape, d_radiance, d_grad_radiance, deltas, 0, stream);
}

int nm_mlp_backward_stops_at_xyz0(const nm_mlp* m) {
    // NM_BACKWARD_STOP_AT_XYZ0 is served by the tuned delta kernels of networks with at least two layers_xyz
    return m && !m->lw && m->plan->generic_nt == 0 && m->desc.num_layers >= 3 && !(getenv("NM_BACKWARD_LINEAR_LAYER1") && atoi(getenv("NM_BACKWARD_LINEAR_LAYER1")) == 0);
}

int nm_mlp_backward_ex(nm_mlp* m, int64_t n, const nm_mlp_tape* tape, const float* d_radiance,
                       const float* d_grad_radiance, const nm_mlp_deltas* deltas, int32_t flags, void* stream) {
    NM_REQUIRE(m && tape && d_radiance && d_grad_radiance && deltas && n >= 0, "bad argument");
    NM_REQUIRE((flags & ~NM_BACKWARD_STOP_AT_XYZ0) == 0, "unknown flag");
    NM_REQUIRE(!(flags & NM_BACKWARD_STOP_AT_XYZ0) || (!m->lw && m->plan->generic_nt == 0 && m->desc.num_layers >= 3),
               "NM_BACKWARD_STOP_AT_XYZ0 needs a tuned-family handle with num_layers >= 3 (ask nm_mlp_backward_stops_at_xyz0)");
    const nm_mlp_desc& d = m->desc;
    const bool flat = d.use_viewdirs == 0;
    const bool generic = m->plan->generic_nt != 0 || m->lw;
    NM_REQUIRE(generic ? (tape->d_h && (flat || (tape->d_feat && tape->d_v))) : (tape->d_mask_h && (flat || tape->d_mask_v)), "incomplete tape");
    NM_REQUIRE(deltas->d_h && deltas->d_last && (flat || (deltas->d_feat && deltas->d_v)), "incomplete delta buffers");
    NM_REQUIRE(m->precision == NM_PREC_F32, "training runs in fp32: create the handle with NM_PREC_F32");
    if (n == 0) return 0;
    if (m->lw) {
        DeviceGuardT guard(m->device);
        return layerwise_backward(m, n, tape, d_radiance, d_grad_radiance, deltas, static_cast<hipStream_t>(stream));
    }
    if (generic) {
        MlpBwdArgs a = m->bwd;
        a.radiance = d_radiance; a.grad_out = d_grad_radiance;
        a.n = n; a.tiles = (n + 15) / 16;
        a.d_h = deltas->d_h; a.d_feat = deltas->d_feat; a.d_v = deltas->d_v; a.d_last = deltas->d_last;
        a.tape_h = tape->d_h; a.tape_feat = tape->d_feat; a.tape_v = tape->d_v;
        const MlpPlan* p = m->plan;
        const int lds_bytes = g_bwd_lds_bytes(p->ring_bytes, p->generic_nt);
        if (int rc = set_lds((const void*)p->kernel_bwd, lds_bytes)) return rc;
        const int64_t wg_iters = (n + p->wg_samples - 1) / p->wg_samples;
        hipLaunchKernelGGL(p->kernel_bwd, dim3(persistent_grid(wg_iters, m->num_cus)), dim3(p->NW * 64), lds_bytes,
                           static_cast<hipStream_t>(stream), a, (int)d.num_layers, flat ? 1 : 0);
        NM_HIP_CHECK(hipGetLastError());
        return 0;
    }
    const BwdPlan* plan = nullptr;
    for (const BwdPlan& p : g_bwd_plans)
        if (p.H == d.hidden_size) plan = &p;
    NM_REQUIRE(plan, "no backward kernel instantiated for this hidden size");
    MlpBwdArgs a = m->bwd;
    a.radiance = d_radiance; a.grad_out = d_grad_radiance;
    a.mask_h = tape->d_mask_h; a.mask_v = tape->d_mask_v;
    a.n = n; a.tiles = (n + 15) / 16;
    a.d_h = deltas->d_h; a.d_feat = deltas->d_feat; a.d_v = deltas->d_v; a.d_last = deltas->d_last;
    a.stop_at_xyz0 = (flags & NM_BACKWARD_STOP_AT_XYZ0) ? 1 : 0;
    const int H = d.hidden_size;
    const int lds_bytes = 2 * KC * (H / 16) * 256 + (((H + (flat ? 3 * H : 3 * H / 2)) * 4 + 255) & ~255);
    const auto kernel = flat ? plan->backward_flat : plan->backward;
    if (int rc = set_lds((const void*)kernel, lds_bytes)) return rc;
    const int64_t wg_iters = (n + 127) / 128;
    hipLaunchKernelGGL(kernel, dim3(persistent_grid(wg_iters, m->num_cus)), dim3(512), lds_bytes,
                       static_cast<hipStream_t>(stream), a, (int)d.num_layers);
    NM_HIP_CHECK(hipGetLastError());
    return 0;
}

int nm_encode_samples_strided(nm_mlp* m, const float* d_origins, int origins_per_ray, const float* d_dirs,
                              const float* d_t, int64_t rays, int32_t samples, float* d_enc_xyz, int32_t stride_xyz,
                              float* d_enc_dir, int32_t stride_dir, void* stream) {
    NM_REQUIRE(m && d_origins && d_dirs && d_t && rays >= 0 && samples > 0, "bad argument");
    EncodeArgs a;
    a.origins = d_origins; a.dirs = d_dirs; a.t = d_t;
    a.n = rays * samples; a.samples = samples; a.origins_per_ray = origins_per_ray;
    a.fx = m->desc.num_encoding_fn_xyz; a.fd = m->desc.num_encoding_fn_dir;
    a.include_x = m->desc.include_input_xyz; a.include_d = m->desc.include_input_dir;
    const int dx = 6 * a.fx + (a.include_x ? 3 : 0), dd = 6 * a.fd + (a.include_d ? 3 : 0);
    NM_REQUIRE((!d_enc_xyz || stride_xyz >= dx) && (!d_enc_dir || stride_dir >= dd), "encode_samples: row stride too small");
    // a handle without view directions keeps no direction bands (nm_mlp_create only range-checks what the kernels use)
    NM_REQUIRE(a.fx <= MAX_FREQ_XYZ && (!d_enc_dir || (a.fd <= MAX_FREQ_DIR && m->desc.use_viewdirs)),
               "encode_samples: this handle has no such encoding (use_viewdirs = 0 handles encode positions only)");
    for (int f = 0; f < MAX_FREQ_XYZ; ++f) a.bands_xyz[f] = m->base.bands_xyz[f];
    for (int f = 0; f < MAX_FREQ_DIR; ++f) a.bands_dir[f] = m->base.bands_dir[f];
    a.enc_x = d_enc_xyz; a.enc_d = d_enc_dir;
    a.stride_x = stride_xyz; a.stride_d = stride_dir;
    if (a.n == 0) return 0;
    // whole rows (zero padding up to the stride included); grid y: the two encodings
    const int items_x = d_enc_xyz ? stride_xyz - 3 * a.fx : 0, items_d = d_enc_dir ? stride_dir - 3 * a.fd : 0;
    const int items = items_x > items_d ? items_x : items_d;
    if (items <= 0) return 0;
    const int narrow = a.n * items + 256 < (1ll << 31) ? 1 : 0;
    const int64_t blocks = (a.n * items + 255) / 256, cap = (int64_t)(m->num_cus > 0 ? m->num_cus : 256) * 8;
    hipLaunchKernelGGL(encode_samples_rows_kernel, dim3((unsigned)(blocks < cap ? blocks : cap), d_enc_dir ? 2 : 1), dim3(256), 0,
                       static_cast<hipStream_t>(stream), a, narrow);
    NM_HIP_CHECK(hipGetLastError());
    return 0;
}

int nm_encode_samples(nm_mlp* m, const float* d_origins, int origins_per_ray, const float* d_dirs, const float* d_t,
                      int64_t rays, int32_t samples, float* d_enc_xyz, float* d_enc_dir, void* stream) {
    NM_REQUIRE(m, "bad argument");
    const int dx = 6 * m->desc.num_encoding_fn_xyz + (m->desc.include_input_xyz ? 3 : 0);
    const int dd = 6 * m->desc.num_encoding_fn_dir + (m->desc.include_input_dir ? 3 : 0);
    return nm_encode_samples_strided(m, d_origins, origins_per_ray, d_dirs, d_t, rays, samples, d_enc_xyz, dx, d_enc_dir, dd,
                                     stream);
}

int nm_mlp_num_cus(const nm_mlp* m) { return m ? m->num_cus : 0; }

}  // extern "C"
