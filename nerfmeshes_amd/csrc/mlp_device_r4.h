// Round-2 experiment (restated for this dataflow): twice the samples per streamed weight byte.
// One wave owns G = 2 groups of 16 samples (32 samples), one wave per SIMD (4 waves / workgroup, 512-register budget);
// every A-operand quad fetched from LDS feeds 4 tiles x G groups = 8 MFMAs instead of 4, which halves the LDS -> VGPR
// operand traffic per FLOP and the number of waves meeting at each barrier.  Weight stream, packer, LDS image, MFMA
// shape (v_mfma_f32_16x16x4_f32) and therefore every fp32 result are those of mlp_kernel / mlp_kernel3: bit-identical.
// Same 3-slot ring / two-chunks-ahead DMA / operand stream across boundaries as mlp_device_r3.h; with no partner wave
// on the SIMD the DMA pieces are issued one at a time inside the MFMA stream.
#pragma once
#include "mlp_device_r3.h"

namespace nm {

template <int G, int NT, int KS1, int KS2, int NW, int LDSBUF, int KCH>
__device__ __forceinline__ void gemm_stage4(f32x4 (&acc)[G][NT], const float (&b1)[G][KS1],
                                            const float (&b2)[G][(KS2 > 0 ? KS2 : 1)], const char* gw,
                                            const NextChunks nx, char* lds, int& slot, f32x4 (&carry)[2],
                                            int wave, int lane) {
    constexpr int KS = KS1 + KS2;
    constexpr int NCH = (KS + KCH - 1) / KCH;
    constexpr int NB = NT / 4;
    constexpr int STEP_BYTES = NT * 256;
    static_assert(NT % 4 == 0 && NB >= 1, "tile count");
    static_assert(NCH >= 2, "every stage must span at least two chunks (DMA runs two chunks ahead)");
    f32x4 ab[3];
    ab[0] = carry[0];
    ab[1] = carry[1];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int steps = (KS - c * KCH) < KCH ? (KS - c * KCH) : KCH;
        const int nblk = steps * NB;
        const int slot1 = slot == 2 ? 0 : slot + 1;
        const int slot2 = slot1 == 2 ? 0 : slot1 + 1;
        const char* src;
        int bytes;
        if (c + 2 < NCH) {
            const int nsteps = (KS - (c + 2) * KCH) < KCH ? (KS - (c + 2) * KCH) : KCH;
            src = gw + (c + 2) * KCH * STEP_BYTES; bytes = nsteps * STEP_BYTES;
        } else if (c + 2 == NCH) { src = nx.s0; bytes = nx.b0; }
        else { src = nx.s1; bytes = nx.b1; }
        char* dst = lds + slot2 * LDSBUF;
        const char* buf = lds + slot * LDSBUF + lane * 16;
        const char* nbuf = lds + slot1 * LDSBUF + lane * 16;
        const int units = (bytes + 1023) >> 10;
        constexpr int MAXP = (LDSBUF / 1024 + NW - 1) / NW;      // DMA pieces one wave may have to issue per chunk
        const int every = nblk / MAXP > 0 ? nblk / MAXP : 1;
#pragma unroll
        for (int j = 0; j < nblk; ++j) {
            if (j % every == 0 && j / every < MAXP) {              // piece (j / every) of this wave, if it exists
                const int u = wave + (j / every) * NW;
                if (u < units)
                    __builtin_amdgcn_global_load_lds(
                        (const __attribute__((address_space(1))) void*)(src + (size_t)u * 1024 + lane * 16),
                        (__attribute__((address_space(3))) void*)(dst + u * 1024), 16, 0, 0);
            }
            const int ks = j / NB, blk = j % NB;
            const int s = c * KCH + ks;
            const char* from = (j + 2 < nblk) ? buf + (j + 2) * 1024 : nbuf + (j + 2 - nblk) * 1024;
            const int r0 = (c * KCH * NB + j) % 3;
            ab[(r0 + 2) % 3] = *reinterpret_cast<const f32x4*>(from);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int gg = 0; gg < G; ++gg) {
                    const float b = s < KS1 ? b1[gg][s < KS1 ? s : 0] : b2[gg][s >= KS1 ? s - KS1 : 0];
                    acc[gg][blk * 4 + q] = __builtin_amdgcn_mfma_f32_16x16x4f32(ab[r0][q], b, acc[gg][blk * 4 + q], 0, 0, 0);
                }
        }
        __syncthreads();
        slot = slot1;
    }
    constexpr int TOTAL = KS * NB;
    carry[0] = ab[TOTAL % 3];
    carry[1] = ab[(TOTAL + 1) % 3];
}

template <int H, int FX, int FD, int NW, int KCH, int G>
__global__ __launch_bounds__(NW * 64, 1) void mlp_kernel4(const MlpArgs args, const int num_layers,
                                                          const int density_only) {
    using N = Net<H, FX, FD, KCH>;
    static_assert(N::EX > KCH && N::KH >= 2 * KCH, "stages must span two chunks");
    extern __shared__ __attribute__((aligned(16))) char lds[];
    float* lds_bias = reinterpret_cast<float*>(lds + 3 * N::LDSBUF);
    const int nbias = H * (1 + num_layers) + H / 2 + 4;
    float* lds_walpha = lds_bias + nbias;
    float* lds_wrgb = lds_walpha + H;
    for (int i = threadIdx.x; i < nbias; i += NW * 64) lds_bias[i] = args.bias[i];
    for (int i = threadIdx.x; i < H; i += NW * 64) lds_walpha[i] = args.walpha[i];
    for (int i = threadIdx.x; i < 3 * H / 2; i += NW * 64) lds_wrgb[i] = args.wrgb[i];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, col = lane & 15;
    const float* tail_bias = lds_bias + nbias - 4;

    constexpr int WG_SAMPLES = NW * 16 * G;
    const int64_t wg_iters = (args.n + WG_SAMPLES - 1) / WG_SAMPLES;
    auto hidden_next = [](const char* p) {
        return NextChunks{p, KCH * N::STEP, p + KCH * N::STEP, (N::KH - KCH < KCH ? N::KH - KCH : KCH) * N::STEP};
    };
    auto enc_next = [](const char* p, bool on) {
        return NextChunks{p, on ? KCH * N::STEP : 0, p + KCH * N::STEP,
                          on ? (N::EX - KCH < KCH ? N::EX - KCH : KCH) * N::STEP : 0};
    };
    auto dir_next = [](const char* p) { return NextChunks{p, KCH * N::STEPD, p + KCH * N::STEPD, KCH * N::STEPD}; };

    int slot = 0;
    f32x4 carry[2];
    if ((int64_t)blockIdx.x < wg_iters) {
        const NextChunks first = enc_next(args.wstream, true);
        stream_to_lds<NW>(first.s0, lds, first.b0, wave, lane);
        stream_to_lds<NW>(first.s1, lds + N::LDSBUF, first.b1, wave, lane);
    }
    __syncthreads();
    carry[0] = *reinterpret_cast<const f32x4*>(lds + lane * 16);
    carry[1] = *reinterpret_cast<const f32x4*>(lds + lane * 16 + 1024);

    for (int64_t it = blockIdx.x; it < wg_iters; it += gridDim.x) {
        const bool has_next = it + gridDim.x < wg_iters;
        int64_t sample[G];
        bool valid[G];
        float d[G][3];
        float encx[G][N::EX];
#pragma unroll
        for (int gg = 0; gg < G; ++gg) {
            sample[gg] = ((it * NW + wave) * G + gg) * 16 + col;
            valid[gg] = sample[gg] < args.n;
            float p[3];
            fetch_sample(args, valid[gg] ? sample[gg] : args.n - 1, p, d[gg]);
            encode<FX, N::EX, 0>(encx[gg], p, args.bands_xyz, g);
        }
        const float dummy[G][1] = {};
        const NextChunks wrap = enc_next(args.wstream, has_next);

        f32x4 acc[G][N::NT];
        float in[G][N::KH];
        const char* gw = args.wstream;
        // ---- layer1: xyz_enc -> H, no activation (models.py:62)
#pragma unroll
        for (int gg = 0; gg < G; ++gg) load_bias<N::NT>(acc[gg], lds_bias, g);
        gemm_stage4<G, N::NT, N::EX, 0, NW, N::LDSBUF, KCH>(acc, encx, dummy, gw, hidden_next(gw + N::EX * N::STEP), lds, slot,
                                                            carry, wave, lane);
        gw += N::EX * N::STEP;
#pragma unroll
        for (int gg = 0; gg < G; ++gg) acc_to_operand<N::NT, false>(acc[gg], in[gg]);

        float sigma[G];
#pragma unroll
        for (int gg = 0; gg < G; ++gg) sigma[gg] = 0.0f;
        const int trunk_iters = density_only ? num_layers - 1 : num_layers;
#pragma unroll 1
        for (int i = 0; i < trunk_iters; ++i) {
            const bool is_feat = i == num_layers - 1;
            if (is_feat) {
#pragma unroll
                for (int gg = 0; gg < G; ++gg) sigma[gg] = alpha_gemv<H>(in[gg], lds_walpha, g) + tail_bias[0];
            }
            const bool skip = !is_feat && ((args.skip_mask >> i) & 1u);
            const bool last_density = density_only && i == num_layers - 2;
#pragma unroll
            for (int gg = 0; gg < G; ++gg) load_bias<N::NT>(acc[gg], lds_bias + H * (1 + i), g);
            {
                const char* after = gw + N::KH * N::STEP;
                NextChunks nx = hidden_next(after);
                if (skip) nx = enc_next(after, true);
                else if (is_feat) nx = dir_next(after);
                else if (last_density) nx = wrap;
                gemm_stage4<G, N::NT, N::KH, 0, NW, N::LDSBUF, KCH>(acc, in, dummy, gw, nx, lds, slot, carry, wave, lane);
                gw = after;
            }
            if (skip) {
                const char* after = gw + N::EX * N::STEP;
                const NextChunks nx = last_density ? wrap : hidden_next(after);
                gemm_stage4<G, N::NT, N::EX, 0, NW, N::LDSBUF, KCH>(acc, encx, dummy, gw, nx, lds, slot, carry, wave, lane);
                gw = after;
            }
#pragma unroll
            for (int gg = 0; gg < G; ++gg) acc_to_operand<N::NT, true>(acc[gg], in[gg]);
        }

        if (density_only) {
#pragma unroll
            for (int gg = 0; gg < G; ++gg) {
                const float sg = alpha_gemv<H>(in[gg], lds_walpha, g) + tail_bias[0];
                if (valid[gg] && g == 0) args.out[sample[gg]] = sg;
            }
            continue;
        }

        // ---- layers_dir[0]: cat(feat, dir_enc) -> H/2, relu (models.py:72-74)
        f32x4 accd[G][N::NTD];
        float v[G][N::KD];
        float encd[G][N::ED];
#pragma unroll
        for (int gg = 0; gg < G; ++gg) {
            load_bias<N::NTD>(accd[gg], lds_bias + H * (1 + num_layers), g);
            encode<FD, N::ED, 0>(encd[gg], d[gg], args.bands_dir, g);
        }
        gemm_stage4<G, N::NTD, N::KH, N::ED, NW, N::LDSBUF, KCH>(accd, in, encd, gw, wrap, lds, slot, carry, wave, lane);
#pragma unroll
        for (int gg = 0; gg < G; ++gg) {
            acc_to_operand<N::NTD, true>(accd[gg], v[gg]);
            float rgb[3];
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                float part = 0.0f;
                const float* wr = lds_wrgb + (ch * 4 + g) * N::KD;
#pragma unroll
                for (int s = 0; s < N::KD; s += 4) {
                    const f32x4 w4 = *reinterpret_cast<const f32x4*>(wr + s);
#pragma unroll
                    for (int q = 0; q < 4; ++q) part = fmaf(v[gg][s + q], w4[q], part);
                }
                const float x = group_sum(part) + tail_bias[1 + ch];
                rgb[ch] = 1.0f / (1.0f + expf(-x));
            }
            if (valid[gg] && g == 0) {
                f32x4 o4 = {rgb[0], rgb[1], rgb[2], sigma[gg]};
                *reinterpret_cast<f32x4*>(args.out + 4 * sample[gg]) = o4;
            }
        }
    }
}

}  // namespace nm
