// Width classes of 25 -- 32 tiles (hidden_size 385 -- 512) of the generic family with a layer's OUTPUT TILES SPLIT OVER TWO WAVES.
//
// Why: a wave of mlp_kernel_g holds a layer's accumulators and its input activation in registers -- 2 * 4 * NT of them --, which
// beyond 24 tiles exceeds the 256 a wave may use at two waves per SIMD; those classes ran 4-wave workgroups, one wave per SIMD
// (0.75 -- 0.83 of the peak against 0.89 -- 0.91 for the classes below: nobody hides the lone wave's LDS and barrier latency).
// Here 16 samples belong to a PAIR of waves.  Each wave of the pair computes half of the output tiles of every layer (all k-steps:
// the same chain order per accumulator as mlp_kernel_g, so the values are bit for bit the one-wave kernel's) and keeps only ITS
// half of the activation in registers; the other half arrives from the partner through LDS, one 16-feature tile per weight chunk:
// a chunk is KCH = 4 k-steps = exactly one input tile, the owner writes tile c + 2 while chunk c runs, the reader fetches tile
// c + 1 after the barrier that ended chunk c - 1 (four 1 KiB slots per pair; D-layout registers ARE the B operand of the next
// layer, so the exchange is lane to same lane: one ds_write_b128 / ds_read_b128 per wave and chunk against 16 operand reads and
// 64 MFMAs).  Registers per wave: 4 * NT accumulators + activations instead of 8 * NT -- 128 at 512 wide -- so the workgroup is 8
// waves, two per SIMD, on the same weight stream, ring, packer and sample order (64 samples per workgroup iteration) as before.
// The 1- and 3-row heads (fc_alpha, fc_rgb / fc_out) are chains over the whole activation: the first wave runs its half of the
// chain, hands the partial sum over, the second continues it (same order of additions), applies the activation and writes the
// output row.  Both encodings are evaluated by both waves (each needs all of their columns for its own output tiles).
#pragma once
#include "mlp_device_g.h"

namespace nm {

constexpr int GS_PAIRS = 4;                       // pairs of waves per workgroup (8 waves)
constexpr int GS_XCH_BYTES = GS_PAIRS * 4 * 1024; // activation exchange: 4 slots of one tile (64 lanes x 16 B) per pair
constexpr int GS_PART_BYTES = GS_PAIRS * 4 * 256; // head partial sums: 4 chains x 64 lanes per pair
constexpr int GS_EXTRA_BYTES = GS_XCH_BYTES + GS_PART_BYTES;   // sits between the weight ring and the biases (counted in the plan's ring_bytes)

// the tiles [T0, T1) of an NT-tile layer that wave `HF` of a pair owns, and the 4-tile operand blocks [B0, B1) they live in
template <int NT, int HF>
struct GsHalf {
    static constexpr int NTA = (NT + 1) / 2;
    static constexpr int T0 = HF ? NTA : 0, T1 = HF ? NT : NTA, N = T1 - T0;
    static constexpr int B0 = T0 / 4, B1 = (T1 + 3) / 4, NBL = B1 - B0;
    static constexpr bool owns(int t) { return t >= T0 && t < T1; }
};

// one chunk's MFMAs of this wave: k-steps `steps` of the chunk resident at `buf` (this lane's 16 B of every block), B values bq
template <int NTO, int HF, int N>
__device__ __forceinline__ void gs_chunk(f32x4 (&acc)[N], const char* buf, const int steps, const float (&bq)[4]) {
    using O = GsHalf<NTO, HF>;
    constexpr int NB = (NTO + 3) / 4;
    const int nblk = steps * O::NBL;              // this wave's blocks of the chunk: (k-step j / NBL, block B0 + j % NBL)
    auto at = [&](int j) { return buf + ((j / O::NBL) * NB + O::B0 + j % O::NBL) * 1024; };
    f32x4 ab[3];
    ab[0] = *reinterpret_cast<const f32x4*>(at(0));
    if (nblk > 1) ab[1] = *reinterpret_cast<const f32x4*>(at(1));
#pragma unroll
    for (int j = 0; j < 4 * O::NBL; ++j) {
        if (j < nblk) {
            if (j + 2 < nblk) ab[(j + 2) % 3] = *reinterpret_cast<const f32x4*>(at(j + 2));
            __builtin_amdgcn_sched_barrier(0);
            const int ks = j / O::NBL, blk = O::B0 + j % O::NBL;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (O::owns(blk * 4 + q))
                    acc[blk * 4 + q - O::T0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ab[j % 3][q], bq[ks], acc[blk * 4 + q - O::T0], 0, 0, 0);
        }
    }
}

// a stage whose B operand is a register array every wave holds whole (an encoding): gemm_stage_g for this wave's tiles
template <int NTO, int HF, int KS, int KCH, bool RUNTIME, int N>
__device__ __forceinline__ void gs_stage_regs(f32x4 (&acc)[N], const float (&b)[KS], int nchunks, const char* gw,
                                              const char* tail_src, int tail_bytes, char* lds, int slot_bytes, int& par,
                                              int wave, int lane) {
    static_assert(KCH == 4 && KS % KCH == 0, "chunks of four k-steps");
    static_assert(N == GsHalf<NTO, HF>::N, "this wave's output tiles");
    constexpr int STEP_BYTES = ((NTO + 3) / 4) * 1024, NCH = KS / KCH;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        if constexpr (RUNTIME) {
            if (c >= nchunks) break;      // uniform
        }
        const bool last = RUNTIME ? (c + 1 >= nchunks) : (c + 1 == NCH);
        stream_to_lds<2 * GS_PAIRS>(last ? tail_src : gw + (c + 1) * KCH * STEP_BYTES, lds + (par ^ 1) * slot_bytes,
                                    last ? tail_bytes : KCH * STEP_BYTES, wave, lane);
        const float bq[4] = {b[4 * c], b[4 * c + 1], b[4 * c + 2], b[4 * c + 3]};
        gs_chunk<NTO, HF>(acc, lds + par * slot_bytes + lane * 16, KCH, bq);
        __syncthreads();
        par ^= 1;
    }
}

// the owner's part of the hand-over before a hidden-input stage: tiles 0 and 1 of the activation into their slots (the caller
// then barriers; whatever else the pair exchanges at this point -- a head's partial sum -- rides on the same barrier)
template <int NTI, int HF, int NI4>
__device__ __forceinline__ void gs_publish_first(const float (&own)[NI4], char* xch, int lane) {
    using I = GsHalf<NTI, HF>;
    static_assert(NI4 == 4 * I::N, "this wave's input tiles");
#pragma unroll
    for (int t = 0; t < 2; ++t)
        if (t < NTI && I::owns(t)) {
            const int i = t - I::T0;
            *reinterpret_cast<f32x4*>(xch + (t % 4) * 1024 + lane * 16) = f32x4{own[4 * i], own[4 * i + 1], own[4 * i + 2], own[4 * i + 3]};
        }
}

// a stage over a hidden activation of NTI tiles split between the pair (chunk c = input tile c): acc[this wave's tiles of NTO] += W x
template <int NTI, int NTO, int HF, int KCH, int N, int NI4>
__device__ __forceinline__ void gs_stage_hidden(f32x4 (&acc)[N], const float (&own)[NI4], char* xch,
                                                const char* gw, const char* tail_src, int tail_bytes, char* lds, int slot_bytes,
                                                int& par, int wave, int lane) {
    static_assert(KCH == 4, "one input tile per chunk");
    using I = GsHalf<NTI, HF>;
    static_assert(N == GsHalf<NTO, HF>::N && NI4 == 4 * I::N, "this wave's tiles");
    constexpr int STEP_BYTES = ((NTO + 3) / 4) * 1024;
    f32x4 xb[2];
    if (!I::owns(0)) xb[0] = *reinterpret_cast<const f32x4*>(xch + lane * 16);
#pragma unroll
    for (int c = 0; c < NTI; ++c) {
        const bool last = c + 1 == NTI;
        stream_to_lds<2 * GS_PAIRS>(last ? tail_src : gw + (c + 1) * KCH * STEP_BYTES, lds + (par ^ 1) * slot_bytes,
                                    last ? tail_bytes : KCH * STEP_BYTES, wave, lane);
        if (c + 2 < NTI && I::owns(c + 2)) {
            const int i = c + 2 - I::T0;
            *reinterpret_cast<f32x4*>(xch + ((c + 2) % 4) * 1024 + lane * 16) = f32x4{own[4 * i], own[4 * i + 1], own[4 * i + 2], own[4 * i + 3]};
        }
        if (c + 1 < NTI && !I::owns(c + 1)) xb[(c + 1) & 1] = *reinterpret_cast<const f32x4*>(xch + ((c + 1) % 4) * 1024 + lane * 16);
        float bq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) bq[q] = I::owns(c) ? own[4 * (I::owns(c) ? c - I::T0 : 0) + q] : xb[c & 1][q];
        gs_chunk<NTO, HF>(acc, lds + par * slot_bytes + lane * 16, KCH, bq);
        __syncthreads();   // DMA landed, slot `par` free, the tile written this chunk visible to the partner
        par ^= 1;
    }
}

template <int NT, int HF, int N>
__device__ __forceinline__ void gs_load_bias(f32x4 (&acc)[N], const float* bias, int g) {
    static_assert(N == GsHalf<NT, HF>::N, "this wave's tiles");
#pragma unroll
    for (int i = 0; i < GsHalf<NT, HF>::N; ++i) acc[i] = *reinterpret_cast<const f32x4*>(bias + 16 * (GsHalf<NT, HF>::T0 + i) + 4 * g);
}

template <int N, bool RELU>
__device__ __forceinline__ void gs_acc_to_own(const f32x4 (&acc)[N], float (&own)[4 * N]) {
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) own[4 * i + r] = RELU ? fmaxf(acc[i][r], 0.0f) : acc[i][r];
}

// this wave's stretch of a head chain (alpha_gemv's order: k-step 4 tile + q ascending), continuing from `start`
template <int NT, int HF, int N4>
__device__ __forceinline__ float gs_chain(const float (&own)[N4], const float* row, float start) {
    static_assert(N4 == 4 * GsHalf<NT, HF>::N, "this wave's tiles");
    float part = start;
#pragma unroll
    for (int i = 0; i < GsHalf<NT, HF>::N; ++i) {
        const f32x4 w4 = *reinterpret_cast<const f32x4*>(row + 4 * (GsHalf<NT, HF>::T0 + i));
#pragma unroll
        for (int q = 0; q < 4; ++q) part = fmaf(own[4 * i + q], w4[q], part);
    }
    return part;
}

// rows of the real width, this wave's tiles only (store_rows_g / load_rows_g / relu_gate of mlp_device_g.h)
template <int NT, int HF, int N4>
__device__ __forceinline__ void gs_store_rows(float* base, int width, int64_t sample, bool valid, const float (&own)[N4], int g) {
    static_assert(N4 == 4 * GsHalf<NT, HF>::N, "this wave's tiles");
    if (!valid) return;
    float* row = base + sample * width;
    const bool vec = (width & 3) == 0;
#pragma unroll
    for (int i = 0; i < GsHalf<NT, HF>::N; ++i) {
        const int k0 = 16 * (GsHalf<NT, HF>::T0 + i) + 4 * g;
        if (vec && k0 + 3 < width) {
            *reinterpret_cast<f32x4*>(row + k0) = f32x4{own[4 * i], own[4 * i + 1], own[4 * i + 2], own[4 * i + 3]};
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (k0 + r < width) row[k0 + r] = own[4 * i + r];
        }
    }
}

// the stages of one encoding's columns for this wave's tiles (enc_stages_g of mlp_device_g.h: one part, or -- LONG -- two)
template <int NTO, int HF, int KCH, bool LONG, int N>
__device__ __forceinline__ void gs_enc_stages(f32x4 (&acc)[N], const float (&x)[3], const GEncArg* tab, int ns, int ident, int ch, int g,
                                              const char* gw, const char* tail_src, int tail_bytes, char* lds, int slot_bytes, int& par,
                                              int wave, int lane) {
    float e[G_ENC_STEPS];
    if constexpr (!LONG) {
        encode_g(e, x, tab, ns, ident, g);
        gs_stage_regs<NTO, HF, G_ENC_STEPS, KCH, true>(acc, e, ch, gw, tail_src, tail_bytes, lds, slot_bytes, par, wave, lane);
    } else {
        constexpr int CPP = G_ENC_STEPS / KCH, STEP_BYTES = ((NTO + 3) / 4) * 1024;
        const int ch0 = ch < CPP ? ch : CPP, ch1 = ch - ch0;                    // uniform
        const char* mid = gw + ch0 * KCH * STEP_BYTES;
        encode_g(e, x, tab, g_part_ns(ns, 0), g_part_ident(ns, ident, 0), g);
        gs_stage_regs<NTO, HF, G_ENC_STEPS, KCH, true>(acc, e, ch0, gw, ch1 ? mid : tail_src, ch1 ? KCH * STEP_BYTES : tail_bytes, lds, slot_bytes,
                                                       par, wave, lane);
        if (ch1) {
            encode_g(e, x, tab + G_ENC_ARGS, g_part_ns(ns, 1), g_part_ident(ns, ident, 1), g);
            gs_stage_regs<NTO, HF, G_ENC_STEPS, KCH, true>(acc, e, ch1, mid, tail_src, tail_bytes, lds, slot_bytes, par, wave, lane);
        }
    }
}

template <int NT, int KCH, bool TAPE, bool LONG, int HF>
__device__ __forceinline__ void gs_forward(const MlpArgs& args, const int num_layers, const int density_only, char* lds,
                                           const float* lds_bias, const int nbias, const float* lds_walpha, const float* lds_wrgb,
                                           const GEncArg* lds_tab, const int wave, const int lane) {
    constexpr int HP = 16 * NT, NTD = (NT + 1) / 2;
    constexpr int KH = 4 * NT, KD = 4 * NTD;
    constexpr int NB = (NT + 3) / 4, NBD = (NTD + 3) / 4;
    constexpr int STEP = NB * 1024, STEPD = NBD * 1024;
    constexpr int SLOT = KCH * STEP;
    constexpr int FIRST_H = KCH * STEP, FIRST_HD = KCH * STEPD;
    using T = GsHalf<NT, HF>;
    using D = GsHalf<NTD, HF>;
    const int pair = wave >> 1;
    char* xch = lds + 2 * SLOT + pair * 4096;
    float* part = reinterpret_cast<float*>(lds + 2 * SLOT + GS_XCH_BYTES + pair * 1024) + lane;    // [chain] at stride 64
    const bool flat = density_only == 2;
    const int g = lane >> 4, col = lane & 15;
    const float* tail_bias = lds_bias + nbias - 4;
    const int chx = args.g_chx, chd = args.g_chd;
    const int enc_x_bytes = chx * KCH * STEP;

    const int64_t wg_iters = (args.n + GS_PAIRS * 16 - 1) / (GS_PAIRS * 16);
    int par = 0;
    for (int64_t it = blockIdx.x; it < wg_iters; it += gridDim.x) {
        const bool has_next = it + gridDim.x < wg_iters;
        const int wrap_bytes = has_next ? KCH * STEP : 0;
        const int64_t sample = (it * GS_PAIRS + pair) * 16 + col;
        const bool valid = sample < args.n;
        const int64_t sidx = valid ? sample : args.n - 1;
        const SamplePD smp = fetch_sample(args, sidx);
        const float p[3] = {smp.px, smp.py, smp.pz}, d[3] = {smp.dx, smp.dy, smp.dz};
        // LONG (16 -- 31 functions, two parts: gs_enc_stages) holds no encoding across the trunk: a skip layer evaluates it again
        // (`opaque` keeps the compiler from hoisting that evaluation back up here)
        auto opaque = [](int v) { asm volatile("" : "+v"(v)); return v; };
        float encx[LONG ? 1 : G_ENC_STEPS];
        if constexpr (!LONG) encode_g(encx, p, lds_tab, args.g_nsx, args.g_idx, g);

        f32x4 acc[T::N];
        float own[4 * T::N];
        const char* gw = args.wstream;
        // ---- layer1: xyz_enc -> H, no activation (models.py:62)
        gs_load_bias<NT, HF>(acc, lds_bias, g);
        if constexpr (!LONG) gs_stage_regs<NT, HF, G_ENC_STEPS, KCH, true>(acc, encx, chx, gw, gw + enc_x_bytes, FIRST_H, lds, SLOT, par, wave, lane);
        else gs_enc_stages<NT, HF, KCH, true>(acc, p, lds_tab, args.g_nsx, args.g_idx, chx, opaque(g), gw, gw + enc_x_bytes, FIRST_H, lds, SLOT, par, wave, lane);
        gw += enc_x_bytes;
        gs_acc_to_own<T::N, false>(acc, own);
        if constexpr (TAPE) gs_store_rows<NT, HF>(args.tape_h, args.g_h, sample, valid, own, g);

        // ---- layers_xyz[0 .. L-2], then (full evaluation only) fc_feat as iteration L-1 (models.py:63-70)
        float sigma = 0.0f;
        const int trunk_iters = density_only ? num_layers - 1 : num_layers;
#pragma unroll 1
        for (int i = 0; i < trunk_iters; ++i) {
            const bool is_feat = i == num_layers - 1;
            // fc_alpha on the pre-feature activation: the first wave's stretch of the chain travels with the first tiles
            if (is_feat && HF == 0) part[0] = gs_chain<NT, HF>(own, lds_walpha + g * (HP / 4), 0.0f);
            gs_publish_first<NT, HF>(own, xch, lane);
            __syncthreads();
            if (is_feat && HF == 1) sigma = group_sum(gs_chain<NT, HF>(own, lds_walpha + g * (HP / 4), part[0])) + tail_bias[0];
            const bool skip = !is_feat && ((args.skip_mask >> i) & 1u);
            const bool last_density = density_only && i == num_layers - 2;
            gs_load_bias<NT, HF>(acc, lds_bias + HP * (1 + i), g);
            {
                const char* after = gw + KH * STEP;
                const char* tsrc = after;
                int tbytes = FIRST_H;
                if (skip) tbytes = KCH * STEP;
                else if (is_feat) tbytes = FIRST_HD;
                else if (last_density) { tsrc = args.wstream; tbytes = wrap_bytes; }
                gs_stage_hidden<NT, NT, HF, KCH>(acc, own, xch, gw, tsrc, tbytes, lds, SLOT, par, wave, lane);
                gw = after;
            }
            if (skip) {   // cat(hidden, xyz_enc): the encoding columns of layers_xyz[i] (models.py:64-65)
                const char* after = gw + enc_x_bytes;
                const char* tsrc = after;
                int tbytes = FIRST_H;
                if (last_density) { tsrc = args.wstream; tbytes = wrap_bytes; }
                if constexpr (!LONG) gs_stage_regs<NT, HF, G_ENC_STEPS, KCH, true>(acc, encx, chx, gw, tsrc, tbytes, lds, SLOT, par, wave, lane);
                else gs_enc_stages<NT, HF, KCH, true>(acc, p, lds_tab, args.g_nsx, args.g_idx, chx, opaque(g), gw, tsrc, tbytes, lds, SLOT, par, wave, lane);
                gw = after;
            }
            gs_acc_to_own<T::N, true>(acc, own);
            if constexpr (TAPE) {
                float* dst = is_feat ? args.tape_feat : args.tape_h + (int64_t)(1 + i) * args.n * args.g_h;
                gs_store_rows<NT, HF>(dst, args.g_h, sample, valid, own, g);
            }
        }

        if (density_only) {     // fc_alpha alone, or (use_viewdirs = 0, models.py:77-79) the four rows of fc_out
            const int rows = flat ? 4 : 1;
            auto row_of = [&](int ch) { return (ch == 0 ? lds_walpha : lds_wrgb + (ch - 1) * HP) + g * (HP / 4); };
            if (HF == 0) {
#pragma unroll
                for (int ch = 0; ch < 4; ++ch)
                    if (ch < rows) part[64 * ch] = gs_chain<NT, HF>(own, row_of(ch), 0.0f);
            }
            __syncthreads();
            if (HF == 1) {
                float x[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int ch = 0; ch < 4; ++ch)
                    if (ch < rows) x[ch] = group_sum(gs_chain<NT, HF>(own, row_of(ch), part[64 * ch])) + tail_bias[ch];
                if (flat) {
                    if (valid && g == 0) {
                        f32x4 o4;
#pragma unroll
                        for (int ch = 0; ch < 3; ++ch) o4[ch] = 1.0f / (1.0f + expf(-x[1 + ch]));
                        o4[3] = x[0];
                        *reinterpret_cast<f32x4*>(args.out + 4 * sample) = o4;
                    }
                } else if (valid && g == 0) args.out[sample] = x[0];
            }
            continue;
        }

        // ---- layers_dir[0]: cat(feat, dir_enc) -> H/2, relu (models.py:72-74): hidden columns, then the encoding columns
        f32x4 accd[D::N];
        float v[4 * D::N];
        gs_publish_first<NT, HF>(own, xch, lane);
        __syncthreads();
        gs_load_bias<NTD, HF>(accd, lds_bias + HP * (1 + num_layers), g);
        {
            const char* after = gw + KH * STEPD;
            const bool has_enc = chd > 0;
            gs_stage_hidden<NT, NTD, HF, KCH>(accd, own, xch, gw, has_enc ? after : args.wstream, has_enc ? KCH * STEPD : wrap_bytes, lds,
                                              SLOT, par, wave, lane);
            gw = after;
            if (has_enc) gs_enc_stages<NTD, HF, KCH, LONG>(accd, d, lds_tab + (LONG ? G_ENC_PARTS : 1) * G_ENC_ARGS, args.g_nsd, args.g_idd, chd, g, gw, args.wstream,
                                                           wrap_bytes, lds, SLOT, par, wave, lane);
        }
        gs_acc_to_own<D::N, true>(accd, v);
        if constexpr (TAPE) gs_store_rows<NTD, HF>(args.tape_v, args.g_hd, sample, valid, v, g);

        // ---- fc_rgb + sigmoid (models.py:75): three chains over the view activation, handed from the first wave to the second
        if (HF == 0) {
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) part[64 * (1 + ch)] = gs_chain<NTD, HF>(v, lds_wrgb + (ch * 4 + g) * KD, 0.0f);
        }
        __syncthreads();
        if (HF == 1) {
            f32x4 o4;
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                const float x = group_sum(gs_chain<NTD, HF>(v, lds_wrgb + (ch * 4 + g) * KD, part[64 * (1 + ch)])) + tail_bias[1 + ch];
                o4[ch] = 1.0f / (1.0f + expf(-x));
            }
            o4[3] = sigma;
            if (valid && g == 0) *reinterpret_cast<f32x4*>(args.out + 4 * sample) = o4;
        }
    }
}

template <int NT, int KCH, bool TAPE = false, bool LONG = false>
__global__ __launch_bounds__(2 * GS_PAIRS * 64, 2) void mlp_kernel_gs(const MlpArgs args, const int num_layers, const int density_only) {
    constexpr int NW = 2 * GS_PAIRS;
    constexpr int HP = 16 * NT, HPD = 16 * ((NT + 1) / 2);
    constexpr int STEP = ((NT + 3) / 4) * 1024, SLOT = KCH * STEP;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    float* lds_bias = reinterpret_cast<float*>(lds + 2 * SLOT + GS_EXTRA_BYTES);
    const int nbias = g_bias_floats(NT, num_layers);
    float* lds_walpha = lds_bias + nbias;
    float* lds_wrgb = lds_walpha + HP;
    const int nrgb = density_only == 2 ? 3 * HP : 3 * HPD;
    GEncArg* lds_tab = reinterpret_cast<GEncArg*>(lds_walpha + g_head_floats(NT));
    for (int i = threadIdx.x; i < nbias; i += NW * 64) lds_bias[i] = args.bias[i];
    for (int i = threadIdx.x; i < HP; i += NW * 64) lds_walpha[i] = args.walpha[i];
    for (int i = threadIdx.x; i < nrgb; i += NW * 64) lds_wrgb[i] = args.wrgb[i];
    for (int i = threadIdx.x; i < 2 * (LONG ? G_ENC_PARTS : 1) * G_ENC_ARGS; i += NW * 64) lds_tab[i] = static_cast<const GEncArg*>(args.g_tab)[i];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t wg_iters = (args.n + GS_PAIRS * 16 - 1) / (GS_PAIRS * 16);
    if ((int64_t)blockIdx.x < wg_iters) stream_to_lds<NW>(args.wstream, lds, KCH * STEP, wave, lane);
    __syncthreads();   // tables, biases and layer1's first chunk are resident
    if (wave & 1) gs_forward<NT, KCH, TAPE, LONG, 1>(args, num_layers, density_only, lds, lds_bias, nbias, lds_walpha, lds_wrgb, lds_tab, wave, lane);
    else gs_forward<NT, KCH, TAPE, LONG, 0>(args, num_layers, density_only, lds, lds_bias, nbias, lds_walpha, lds_wrgb, lds_tab, wave, lane);
}

// ---- delta propagation (mlp_backward_kernel_g of mlp_device_g.h) on the same split: the transposed layers in reverse order, each
// wave of a pair producing its half of a delta's tiles, gating them with ITS tiles of the taped activation and storing ITS part of
// every delta row; the other half of the delta reaches the next stage through the exchange slots as in the forward kernel.
template <int NT, int HF, int N, int N4>
__device__ __forceinline__ void gs_relu_gate(const f32x4 (&acc)[N], const float* base, int width, int64_t sample, float (&own)[N4], int g) {
    static_assert(N == GsHalf<NT, HF>::N && N4 == 4 * N, "this wave's tiles");
    const float* row = base + sample * width;
    const bool vec = (width & 3) == 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const int k0 = 16 * (GsHalf<NT, HF>::T0 + i) + 4 * g;
        f32x4 a = {0.0f, 0.0f, 0.0f, 0.0f};
        if (vec && k0 + 3 < width) a = *reinterpret_cast<const f32x4*>(row + k0);
        else {
#pragma unroll
            for (int r = 0; r < 4; ++r) a[r] = k0 + r < width ? row[k0 + r] : 0.0f;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) own[4 * i + r] = a[r] > 0.0f ? acc[i][r] : 0.0f;
    }
}

template <int NT, int KCH, int HF>
__device__ __forceinline__ void gs_backward(const MlpBwdArgs& args, const int num_layers, const int flat, char* lds,
                                            const float* lds_walpha, const float* lds_wrgb, const int wave, const int lane) {
    constexpr int HP = 16 * NT, NTD = (NT + 1) / 2;
    constexpr int KH = 4 * NT, KD = 4 * NTD;
    constexpr int STEP = ((NT + 3) / 4) * 1024, SLOT = KCH * STEP;
    constexpr int FIRST = KCH * STEP;                           // first chunk of any stage (every stage lands on the trunk tiles)
    using T = GsHalf<NT, HF>;
    using D = GsHalf<NTD, HF>;
    const int pair = wave >> 1;
    char* xch = lds + 2 * SLOT + pair * 4096;
    const int g = lane >> 4, col = lane & 15;
    const int L = num_layers, H = args.g_h, HD = args.g_hd;

    const int64_t wg_iters = (args.n + GS_PAIRS * 16 - 1) / (GS_PAIRS * 16);
    int par = 0;
    for (int64_t it = blockIdx.x; it < wg_iters; it += gridDim.x) {
        const bool has_next = it + gridDim.x < wg_iters;
        const int wrap_bytes = has_next ? FIRST : 0;
        const int64_t sample = (it * GS_PAIRS + pair) * 16 + col;
        const bool valid = sample < args.n;
        const int64_t sidx = valid ? sample : args.n - 1;
        // ---- head: sigmoid' (models.py:75 / :78)
        const f32x4 go = *reinterpret_cast<const f32x4*>(args.grad_out + 4 * sidx);
        const f32x4 y = *reinterpret_cast<const f32x4*>(args.radiance + 4 * sidx);
        float drgb[3];
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) drgb[ch] = go[ch] * (y[ch] * (1.0f - y[ch]));
        const float dsigma = go[3];
        if (HF == 0 && valid && g == 0) *reinterpret_cast<f32x4*>(args.d_last + 4 * sample) = f32x4{drgb[0], drgb[1], drgb[2], dsigma};
        f32x4 acc[T::N];
        float own[4 * T::N];
        const char* gw = args.wstream;
        const float* wa = lds_walpha + g * (HP / 4);
        if (flat) {
            // ---- fc_out^T on the VALU: delta at the trunk's output from the four head deltas (rows in fc_alpha's operand layout)
#pragma unroll
            for (int i = 0; i < T::N; ++i) {
                const int nt = T::T0 + i;
                f32x4 w4 = *reinterpret_cast<const f32x4*>(wa + 4 * nt);
                f32x4 a = {w4[0] * dsigma, w4[1] * dsigma, w4[2] * dsigma, w4[3] * dsigma};
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) {
                    w4 = *reinterpret_cast<const f32x4*>(lds_wrgb + ch * HP + g * (HP / 4) + 4 * nt);
#pragma unroll
                    for (int r = 0; r < 4; ++r) a[r] = fmaf(w4[r], drgb[ch], a[r]);
                }
                acc[i] = a;
            }
        } else {
            // ---- fc_rgb^T on the VALU, gated by the view layer's ReLU: this wave's tiles of the delta at layers_dir.0's pre-activation
            float dv[4 * D::N];
            {
                const float* row = args.tape_v + sidx * HD;
                const bool vec = (HD & 3) == 0;
#pragma unroll
                for (int i = 0; i < D::N; ++i) {
                    const int k0 = 16 * (D::T0 + i) + 4 * g;
                    f32x4 av = {0.0f, 0.0f, 0.0f, 0.0f};
                    if (vec && k0 + 3 < HD) av = *reinterpret_cast<const f32x4*>(row + k0);
                    else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) av[r] = k0 + r < HD ? row[k0 + r] : 0.0f;
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int s = 4 * (D::T0 + i) + r;
                        float a = lds_wrgb[(0 * 4 + g) * KD + s] * drgb[0];
                        a = fmaf(lds_wrgb[(1 * 4 + g) * KD + s], drgb[1], a);
                        a = fmaf(lds_wrgb[(2 * 4 + g) * KD + s], drgb[2], a);
                        dv[4 * i + r] = av[r] > 0.0f ? a : 0.0f;
                    }
                }
            }
            gs_store_rows<NTD, HF>(args.d_v, HD, sample, valid, dv, g);
            // ---- layers_dir.0^T (hidden columns): -> delta at relu(fc_feat) -> gated -> delta at fc_feat's output
#pragma unroll
            for (int i = 0; i < T::N; ++i) acc[i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            gs_publish_first<NTD, HF>(dv, xch, lane);
            __syncthreads();
            gs_stage_hidden<NTD, NT, HF, KCH>(acc, dv, xch, gw, gw + KD * STEP, FIRST, lds, SLOT, par, wave, lane);
            gw += KD * STEP;
            gs_relu_gate<NT, HF>(acc, args.tape_feat, H, sidx, own, g);
            gs_store_rows<NT, HF>(args.d_feat, H, sample, valid, own, g);
            // ---- fc_feat^T + fc_alpha^T: delta at the output of layers_xyz[L-2]
#pragma unroll
            for (int i = 0; i < T::N; ++i) {
                const f32x4 w4 = *reinterpret_cast<const f32x4*>(wa + 4 * (T::T0 + i));
                acc[i] = f32x4{w4[0] * dsigma, w4[1] * dsigma, w4[2] * dsigma, w4[3] * dsigma};
            }
            gs_publish_first<NT, HF>(own, xch, lane);
            __syncthreads();
            gs_stage_hidden<NT, NT, HF, KCH>(acc, own, xch, gw, gw + KH * STEP, FIRST, lds, SLOT, par, wave, lane);
            gw += KH * STEP;
        }
        gs_relu_gate<NT, HF>(acc, args.tape_h + (int64_t)(L - 1) * args.n * H, H, sidx, own, g);
        gs_store_rows<NT, HF>(args.d_h + (int64_t)(L - 1) * args.n * H, H, sample, valid, own, g);
        // ---- layers_xyz[i]^T, i = L-2 .. 0: delta at the input of layers_xyz[i] (gated by the ReLU of layers_xyz[i-1]; layer1 has none)
#pragma unroll 1
        for (int i = L - 2; i >= 0; --i) {
#pragma unroll
            for (int t = 0; t < T::N; ++t) acc[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            const char* after = gw + KH * STEP;
            gs_publish_first<NT, HF>(own, xch, lane);
            __syncthreads();
            gs_stage_hidden<NT, NT, HF, KCH>(acc, own, xch, gw, i == 0 ? args.wstream : after, i == 0 ? wrap_bytes : FIRST, lds, SLOT, par,
                                             wave, lane);
            gw = after;
            if (i > 0) gs_relu_gate<NT, HF>(acc, args.tape_h + (int64_t)i * args.n * H, H, sidx, own, g);
            else gs_acc_to_own<T::N, false>(acc, own);
            gs_store_rows<NT, HF>(args.d_h + (int64_t)i * args.n * H, H, sample, valid, own, g);
        }
    }
}

template <int NT, int KCH>
__global__ __launch_bounds__(2 * GS_PAIRS * 64, 2) void mlp_backward_kernel_gs(const MlpBwdArgs args, const int num_layers, const int flat) {
    constexpr int NW = 2 * GS_PAIRS;
    constexpr int HP = 16 * NT, HPD = 16 * ((NT + 1) / 2);
    constexpr int STEP = ((NT + 3) / 4) * 1024, SLOT = KCH * STEP;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    float* lds_walpha = reinterpret_cast<float*>(lds + 2 * SLOT + GS_EXTRA_BYTES);   // [4][HP / 4]
    float* lds_wrgb = lds_walpha + HP;                                               // [3][4][HPD / 4], or (flat) [3][4][HP / 4]
    for (int i = threadIdx.x; i < HP; i += NW * 64) lds_walpha[i] = args.walpha[i];
    for (int i = threadIdx.x; i < (flat ? 3 * HP : 3 * HPD); i += NW * 64) lds_wrgb[i] = args.wrgb[i];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t wg_iters = (args.n + GS_PAIRS * 16 - 1) / (GS_PAIRS * 16);
    if ((int64_t)blockIdx.x < wg_iters) stream_to_lds<NW>(args.wstream, lds, KCH * STEP, wave, lane);
    __syncthreads();
    if (wave & 1) gs_backward<NT, KCH, 1>(args, num_layers, flat, lds, lds_walpha, lds_wrgb, wave, lane);
    else gs_backward<NT, KCH, 0>(args, num_layers, flat, lds, lds_walpha, lds_wrgb, wave, lane);
}

}  // namespace nm
