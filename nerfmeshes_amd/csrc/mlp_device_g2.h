// EXPERIMENT (ablation library only, -DNM_ABLATIONS): the generic forward kernel with TWO 16-sample column tiles per wave
// ("give each wave two column tiles for H <= 128: one A-operand ds_read_b128 then feeds 8 MFMAs").
// Same values as mlp_kernel_g (same k-order per accumulator); what changes is the work per LDS read / DMA piece / barrier
// (halved per FLOP) and the registers per lane (doubled).  View-dependent and density-only evaluation; measured by
// tests/tools/bench_ct2.py against the tuned and the one-tile generic kernels (profiles/r04_two_column_tiles.json).
#pragma once
#include <type_traits>

#include "mlp_device_g.h"

namespace nm {

// both column tiles with a COMPILE-TIME tile index (a `for (ct ...)` over bodies with control flow is not always unrolled,
// and a run-time index would put the per-tile register arrays into scratch)
#define NM_BOTH_TILES(...) do { auto nm_tile_ = [&](auto CT_) { constexpr int ct = decltype(CT_)::value; __VA_ARGS__ }; \
        nm_tile_(std::integral_constant<int, 0>{}); nm_tile_(std::integral_constant<int, 1>{}); } while (0)

template <int NT, int KS, int NW, int KCH, bool RUNTIME>
__device__ __forceinline__ void gemm_stage_g2(f32x4 (&acc)[2][NT], const float (&b)[2][KS], int nchunks, const char* gw,
                                              const char* tail_src, int tail_bytes, char* lds, int slot_bytes, int& par,
                                              int wave, int lane) {
    constexpr int NB = (NT + 3) / 4;
    constexpr int STEP_BYTES = NB * 1024;
    constexpr int NCH = (KS + KCH - 1) / KCH;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        if constexpr (RUNTIME) {
            if (c >= nchunks) break;
        }
        const int steps = (KS - c * KCH) < KCH ? (KS - c * KCH) : KCH;
        const int steps_next = (KS - (c + 1) * KCH) < KCH ? (KS - (c + 1) * KCH) : KCH;
        const bool last = RUNTIME ? (c + 1 >= nchunks) : (c + 1 == NCH);
        const char* next_src = last ? tail_src : gw + (c + 1) * KCH * STEP_BYTES;
        const int next_bytes = last ? tail_bytes : steps_next * STEP_BYTES;
        stream_to_lds<NW>(next_src, lds + (par ^ 1) * slot_bytes, next_bytes, wave, lane);
        const char* buf = lds + par * slot_bytes + lane * 16;
        const int nblk = steps * NB;
        f32x4 ab[3];
        ab[0] = *reinterpret_cast<const f32x4*>(buf);
        if (nblk > 1) ab[1] = *reinterpret_cast<const f32x4*>(buf + 1024);
#pragma unroll
        for (int j = 0; j < nblk; ++j) {
            if (j + 2 < nblk) ab[(j + 2) % 3] = *reinterpret_cast<const f32x4*>(buf + (j + 2) * 1024);
            __builtin_amdgcn_sched_barrier(0);
            const int ks = j / NB, blk = j % NB;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (blk * 4 + q < NT) {
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct)
                        acc[ct][blk * 4 + q] = __builtin_amdgcn_mfma_f32_16x16x4f32(ab[j % 3][q], b[ct][c * KCH + ks], acc[ct][blk * 4 + q], 0, 0, 0);
                }
        }
        __syncthreads();
        par ^= 1;
    }
}

// fc_rgb + sigmoid (models.py:75): 3-row GEMV on the VALU, as in mlp_kernel_g
template <int KD>
__device__ __forceinline__ void rgb_head_g2(const float (&v)[KD], const float* lds_wrgb, const float* tail_bias, int g, float (&rgb)[3]) {
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        float part = 0.0f;
        const float* wr = lds_wrgb + (ch * 4 + g) * KD;
#pragma unroll
        for (int s = 0; s < KD; s += 4) {
            const f32x4 w4 = *reinterpret_cast<const f32x4*>(wr + s);
#pragma unroll
            for (int q = 0; q < 4; ++q) part = fmaf(v[s + q], w4[q], part);
        }
        const float x = group_sum(part) + tail_bias[1 + ch];
        rgb[ch] = 1.0f / (1.0f + expf(-x));
    }
}

template <int NT, int NW, int KCH, int WPS = 2>
__global__ __launch_bounds__(NW * 64, WPS) void mlp_kernel_g2(const MlpArgs args, const int num_layers, const int density_only) {
    constexpr int HP = 16 * NT, NTD = (NT + 1) / 2, HPD = 16 * NTD;
    constexpr int KH = 4 * NT, KD = 4 * NTD;
    constexpr int NB = (NT + 3) / 4, NBD = (NTD + 3) / 4;
    constexpr int STEP = NB * 1024, STEPD = NBD * 1024;
    constexpr int SLOT = KCH * STEP;
    constexpr int FIRST_H = (KH < KCH ? KH : KCH) * STEP, FIRST_HD = (KH < KCH ? KH : KCH) * STEPD;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    float* lds_bias = reinterpret_cast<float*>(lds + 2 * SLOT);
    const int nbias = HP * (1 + num_layers) + HPD + 4;
    float* lds_walpha = lds_bias + nbias;
    float* lds_wrgb = lds_walpha + HP;
    GEncArg* lds_tab = reinterpret_cast<GEncArg*>(lds_wrgb + 3 * HP);
    for (int i = threadIdx.x; i < nbias; i += NW * 64) lds_bias[i] = args.bias[i];
    for (int i = threadIdx.x; i < HP; i += NW * 64) lds_walpha[i] = args.walpha[i];
    for (int i = threadIdx.x; i < 3 * HPD; i += NW * 64) lds_wrgb[i] = args.wrgb[i];
    for (int i = threadIdx.x; i < 2 * G_ENC_ARGS; i += NW * 64) lds_tab[i] = static_cast<const GEncArg*>(args.g_tab)[i];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, col = lane & 15;
    const float* tail_bias = lds_bias + nbias - 4;
    const int chx = args.g_chx, chd = args.g_chd;
    const int enc_x_bytes = chx * KCH * STEP;

    const int64_t wg_iters = (args.n + NW * 32 - 1) / (NW * 32);
    int par = 0;
    if ((int64_t)blockIdx.x < wg_iters) stream_to_lds<NW>(args.wstream, lds, KCH * STEP, wave, lane);
    __syncthreads();

    for (int64_t it = blockIdx.x; it < wg_iters; it += gridDim.x) {
        const bool has_next = it + gridDim.x < wg_iters;
        const int wrap_bytes = has_next ? KCH * STEP : 0;
        int64_t sample[2];
        bool valid[2];
        float encx[2][G_ENC_STEPS], dir[2][3];
        NM_BOTH_TILES(
            sample[ct] = ((it * NW + wave) * 2 + ct) * 16 + col;
            valid[ct] = sample[ct] < args.n;
            const SamplePD smp = fetch_sample(args, valid[ct] ? sample[ct] : args.n - 1);
            const float p[3] = {smp.px, smp.py, smp.pz};
            dir[ct][0] = smp.dx; dir[ct][1] = smp.dy; dir[ct][2] = smp.dz;
            encode_g(encx[ct], p, lds_tab, args.g_nsx, args.g_idx, g);
        );
        f32x4 acc[2][NT];
        float in[2][KH];
        const char* gw = args.wstream;
        NM_BOTH_TILES(load_bias<NT>(acc[ct], lds_bias, g););
        gemm_stage_g2<NT, G_ENC_STEPS, NW, KCH, true>(acc, encx, chx, gw, gw + enc_x_bytes, FIRST_H, lds, SLOT, par, wave, lane);
        gw += enc_x_bytes;
        NM_BOTH_TILES(acc_to_operand<NT, false>(acc[ct], in[ct]););

        float sigma[2] = {0.0f, 0.0f};
        const int trunk_iters = density_only ? num_layers - 1 : num_layers;
#pragma unroll 1
        for (int i = 0; i < trunk_iters; ++i) {
            const bool is_feat = i == num_layers - 1;
            if (is_feat) NM_BOTH_TILES(sigma[ct] = alpha_gemv<HP>(in[ct], lds_walpha, g) + tail_bias[0];);
            const bool skip = !is_feat && ((args.skip_mask >> i) & 1u);
            const bool last_density = density_only && i == num_layers - 2;
            NM_BOTH_TILES(load_bias<NT>(acc[ct], lds_bias + HP * (1 + i), g););
            {
                const char* after = gw + KH * STEP;
                const char* tsrc = after;
                int tbytes = FIRST_H;
                if (skip) tbytes = KCH * STEP;
                else if (is_feat) tbytes = FIRST_HD;
                else if (last_density) { tsrc = args.wstream; tbytes = wrap_bytes; }
                gemm_stage_g2<NT, KH, NW, KCH, false>(acc, in, 0, gw, tsrc, tbytes, lds, SLOT, par, wave, lane);
                gw = after;
            }
            if (skip) {
                const char* after = gw + enc_x_bytes;
                const char* tsrc = after;
                int tbytes = FIRST_H;
                if (last_density) { tsrc = args.wstream; tbytes = wrap_bytes; }
                gemm_stage_g2<NT, G_ENC_STEPS, NW, KCH, true>(acc, encx, chx, gw, tsrc, tbytes, lds, SLOT, par, wave, lane);
                gw = after;
            }
            NM_BOTH_TILES(acc_to_operand<NT, true>(acc[ct], in[ct]););
        }
        if (density_only) {
            NM_BOTH_TILES(
                const float s = alpha_gemv<HP>(in[ct], lds_walpha, g) + tail_bias[0];
                if (valid[ct] && g == 0) args.out[sample[ct]] = s;
            );
            continue;
        }
        f32x4 accd[2][NTD];
        float v[2][KD];
        NM_BOTH_TILES(load_bias<NTD>(accd[ct], lds_bias + HP * (1 + num_layers), g););
        {
            const char* after = gw + KH * STEPD;
            const bool has_enc = chd > 0;
            gemm_stage_g2<NTD, KH, NW, KCH, false>(accd, in, 0, gw, has_enc ? after : args.wstream,
                                                   has_enc ? KCH * STEPD : wrap_bytes, lds, SLOT, par, wave, lane);
            gw = after;
            if (has_enc) {
                float encd[2][G_ENC_STEPS];
                NM_BOTH_TILES(encode_g(encd[ct], dir[ct], lds_tab + G_ENC_ARGS, args.g_nsd, args.g_idd, g););
                gemm_stage_g2<NTD, G_ENC_STEPS, NW, KCH, true>(accd, encd, chd, gw, args.wstream, wrap_bytes, lds, SLOT, par, wave, lane);
            }
        }
        NM_BOTH_TILES(
            acc_to_operand<NTD, true>(accd[ct], v[ct]);
            float rgb[3];
            rgb_head_g2<KD>(v[ct], lds_wrgb, tail_bias, g, rgb);
            if (valid[ct] && g == 0) {
                f32x4 o4 = {rgb[0], rgb[1], rgb[2], sigma[ct]};
                *reinterpret_cast<f32x4*>(args.out + 4 * sample[ct]) = o4;
            }
        );
    }
}

}  // namespace nm
