// EXPERIMENT, ablation library only (#ifdef NM_ABLATIONS in nerf_mlp.hip; NM_MLP_VARIANT=300 selects it for bf16x3 handles):
// the bf16x3 kernel with TWO 16-sample column tiles per wave.  Measured (profiles/r04_bf16x3_two_tiles.json):
// bit-identical to mlp_kernel_b3, 252 TFLOP/s fp32-equivalent against 274 -- not adopted.  Why: per 32 samples the kernel
// issues 12.2 k non-matrix VALU instructions (operand conversion 5.5 per value, AGPR moves, encodings) next to 13.9 k MFMAs;
// with 481 registers a SIMD holds ONE wave, and a lone wave issues in order: its VALU work runs while the matrix pipe
// idles (PMC: pipe 61.5 % busy, 70.3 % for the two-waves-per-SIMD kernel, whose LDS traffic is twice as high but whose
// partner wave fills those gaps).
#pragma once
#include <type_traits>
#include "mlp_device_b3.h"

namespace nm {
// ---- two 16-sample column tiles per wave ("w"): one 3 KiB operand fetch feeds 12 matrix instructions -------------------
// The single-tile kernel above runs at the LDS peak: 3 planes (3 KiB per wave) per 6 MFMAs = 128 B/clk per CU at the full
// matrix rate.  Here a wave owns 32 samples: the A planes of a unit are read once and used for both tiles (64 B/clk), at
// the price of 2 x (64 accumulator + 96 split-operand) registers per lane, i.e. ONE wave per SIMD (NW = 4 waves of 32
// samples = the same 128-sample workgroup, the same weight stream, ring and DMA schedule).  With a single wave per SIMD
// nothing but this wave's own instruction order hides latency: the two tiles' MFMA chains are interleaved (a dependent
// pair is always one independent instruction apart), the A planes are fetched two units ahead as before.  Per sample the
// arithmetic is that of the single-tile kernel, instruction for instruction: the results are the same bits.
template <class F>
__device__ __forceinline__ void b3_both(F&& f) {
    f(std::integral_constant<int, 0>{});
    f(std::integral_constant<int, 1>{});
}

template <int NT, int KB1, int KB2, int NW>
__device__ __forceinline__ void gemm_stage_b3w(f32x4 (&acc0)[NT], f32x4 (&acc1)[NT], const Split3 (&b10)[KB1], const Split3 (&b11)[KB1],
                                               const Split3 (&b20)[(KB2 > 0 ? KB2 : 1)], const Split3 (&b21)[(KB2 > 0 ? KB2 : 1)],
                                               const char* gw, const B3Next nx, char* lds, int& slot, u32x4 (&carry)[2][3],
                                               int wave, int lane) {
    constexpr int UNITS = (KB1 + KB2) * NT;
    constexpr int NCH = (UNITS + B3_CHUNK_UNITS - 1) / B3_CHUNK_UNITS;
    static_assert(NCH >= 2, "every stage must span at least two chunks");
    u32x4 ab[3][3];
#pragma unroll
    for (int p = 0; p < 3; ++p) { ab[0][p] = carry[0][p]; ab[1][p] = carry[1][p]; }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int units = (UNITS - c * B3_CHUNK_UNITS) < B3_CHUNK_UNITS ? (UNITS - c * B3_CHUNK_UNITS) : B3_CHUNK_UNITS;
        const int slot1 = slot == 2 ? 0 : slot + 1;
        const int slot2 = slot1 == 2 ? 0 : slot1 + 1;
        const char* src;
        int bytes;
        if (c + 2 < NCH) {
            const int nu = (UNITS - (c + 2) * B3_CHUNK_UNITS) < B3_CHUNK_UNITS ? (UNITS - (c + 2) * B3_CHUNK_UNITS) : B3_CHUNK_UNITS;
            src = gw + (c + 2) * B3_SLOT; bytes = nu * B3_UNIT;
        } else if (c + 2 == NCH) { src = nx.s0; bytes = nx.b0; }
        else { src = nx.s1; bytes = nx.b1; }
        char* dst = lds + slot2 * B3_SLOT;
        const char* buf = lds + slot * B3_SLOT + lane * 16;
        const char* nbuf = lds + slot1 * B3_SLOT + lane * 16;
#pragma unroll
        for (int j = 0; j < units; ++j) {
            if (j == 0 && wave < NW / 2) stream_to_lds<NW>(src, dst, bytes, wave, lane);
            if (j == units / 2 && wave >= NW / 2) stream_to_lds<NW>(src, dst, bytes, wave, lane);
            const int u = c * B3_CHUNK_UNITS + j;
            const int m = u / NT, nt = u % NT;
            const char* from = (j + 2 < units) ? buf + (j + 2) * B3_UNIT : nbuf + (j + 2 - units) * B3_UNIT;
            const int r0 = u % 3;
#pragma unroll
            for (int p = 0; p < 3; ++p) ab[(r0 + 2) % 3][p] = *reinterpret_cast<const u32x4*>(from + p * 1024);
            __builtin_amdgcn_sched_barrier(0);
            const Split3& s0 = m < KB1 ? b10[m < KB1 ? m : 0] : b20[m >= KB1 ? m - KB1 : 0];
            const Split3& s1 = m < KB1 ? b11[m < KB1 ? m : 0] : b21[m >= KB1 ? m - KB1 : 0];
            const bf16x8 a1 = __builtin_bit_cast(bf16x8, ab[r0][0]), a2 = __builtin_bit_cast(bf16x8, ab[r0][1]),
                         a3 = __builtin_bit_cast(bf16x8, ab[r0][2]);
            const bf16x8 x1 = __builtin_bit_cast(bf16x8, s0.p[0]), x2 = __builtin_bit_cast(bf16x8, s0.p[1]),
                         x3 = __builtin_bit_cast(bf16x8, s0.p[2]);
            const bf16x8 y1 = __builtin_bit_cast(bf16x8, s1.p[0]), y2 = __builtin_bit_cast(bf16x8, s1.p[1]),
                         y3 = __builtin_bit_cast(bf16x8, s1.p[2]);
            f32x4 d = acc0[nt], e = acc1[nt];
            d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3, x1, d, 0, 0, 0);   // smallest terms first, per tile as above
            e = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3, y1, e, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, x2, d, 0, 0, 0);
            e = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, y2, e, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, x3, d, 0, 0, 0);
            e = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, y3, e, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, x1, d, 0, 0, 0);
            e = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, y1, e, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, x2, d, 0, 0, 0);
            e = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, y2, e, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, x1, d, 0, 0, 0);
            e = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, y1, e, 0, 0, 0);
            acc0[nt] = d; acc1[nt] = e;
        }
        __syncthreads();
        slot = slot1;
    }
#pragma unroll
    for (int p = 0; p < 3; ++p) { carry[0][p] = ab[UNITS % 3][p]; carry[1][p] = ab[(UNITS + 1) % 3][p]; }
}

// fc_rgb + sigmoid of one tile from the fp32 accumulators of the view layer
template <int H, int NTD>
__device__ __forceinline__ void b3_rgb_head(const f32x4 (&accd)[NTD], const float* lds_wrgb, const float* tail_bias, int g, float (&rgb)[3]) {
    float v[4 * NTD];
    acc_to_operand<NTD, true>(accd, v);
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        float part = 0.0f;
        const float* wr = lds_wrgb + (ch * 4 + g) * (H / 8);
#pragma unroll
        for (int s = 0; s < H / 8; s += 4) {
            const f32x4 w4 = *reinterpret_cast<const f32x4*>(wr + s);
#pragma unroll
            for (int q = 0; q < 4; ++q) part = fmaf(v[s + q], w4[q], part);
        }
        const float x = group_sum(part) + tail_bias[1 + ch];
        rgb[ch] = 1.0f / (1.0f + expf(-x));
    }
}

template <int H, int FX, int FD, int NW>
__global__ __launch_bounds__(NW * 64, 1) void mlp_kernel_b3w(const MlpArgs args, const int num_layers, const int density_only) {
    constexpr int NT = H / 16, KB = H / 32, NTD = H / 32, KBX = 2, KBD = 1;
    static_assert(6 * FX + 3 <= 64 && 6 * FD + 3 <= 32 && FX <= 16 && FD <= 16, "encoding slots");
    static_assert(NT * KBX >= 2 * B3_CHUNK_UNITS && NTD * (KB + KBD) >= 2 * B3_CHUNK_UNITS, "stages must span two chunks");
    extern __shared__ __attribute__((aligned(16))) char lds[];
    float* lds_bias = reinterpret_cast<float*>(lds + 3 * B3_SLOT);
    const int nbias = H * (1 + num_layers) + H / 2 + 4;
    float* lds_walpha = lds_bias + nbias;
    float* lds_wrgb = lds_walpha + H;
    for (int i = threadIdx.x; i < nbias; i += NW * 64) lds_bias[i] = args.bias[i];
    for (int i = threadIdx.x; i < H; i += NW * 64) lds_walpha[i] = args.walpha[i];
    for (int i = threadIdx.x; i < 3 * H / 2; i += NW * 64) lds_wrgb[i] = args.wrgb[i];
    float* lds_bands = lds_wrgb + 3 * H / 2;
    if (threadIdx.x < FX) lds_bands[threadIdx.x] = args.bands_xyz[threadIdx.x];
    if (threadIdx.x < FD) lds_bands[16 + threadIdx.x] = args.bands_dir[threadIdx.x];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, col = lane & 15;
    const float* tail_bias = lds_bias + nbias - 4;

    constexpr int U_ENC = KBX * NT, U_HID = KB * NT, U_DIR = (KB + KBD) * NTD;
    auto next_of = [](const char* p, int units, bool on) {
        const int u0 = units < B3_CHUNK_UNITS ? units : B3_CHUNK_UNITS;
        const int u1 = units - u0 < B3_CHUNK_UNITS ? units - u0 : B3_CHUNK_UNITS;
        return B3Next{p, on ? u0 * B3_UNIT : 0, p + B3_SLOT, on ? u1 * B3_UNIT : 0};
    };
    const int64_t wg_iters = (args.n + NW * 32 - 1) / (NW * 32);
    int slot = 0;
    u32x4 carry[2][3];
    if ((int64_t)blockIdx.x < wg_iters) {
        const B3Next first = next_of(args.wstream, U_ENC, true);
        stream_to_lds<NW>(first.s0, lds, first.b0, wave, lane);
        stream_to_lds<NW>(first.s1, lds + B3_SLOT, first.b1, wave, lane);
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int p = 0; p < 3; ++p) carry[u][p] = *reinterpret_cast<const u32x4*>(lds + u * B3_UNIT + p * 1024 + lane * 16);

    for (int64_t it = blockIdx.x; it < wg_iters; it += gridDim.x) {
        const bool has_next = it + gridDim.x < wg_iters;
        auto opaque = [](int v) { asm volatile("" : "+v"(v)); return v; };
        int64_t sample[2];
        bool valid[2];
        float p[2][3], d[2][3];
        b3_both([&](auto ct) __attribute__((always_inline)) {
            sample[ct] = ((it * NW + wave) * 2 + ct) * 16 + col;
            valid[ct] = sample[ct] < args.n;
            const SamplePD smp = fetch_sample(args, valid[ct] ? sample[ct] : args.n - 1);
            p[ct][0] = smp.px; p[ct][1] = smp.py; p[ct][2] = smp.pz;
            d[ct][0] = smp.dx; d[ct][1] = smp.dy; d[ct][2] = smp.dz;
        });
        const B3Next wrap = next_of(args.wstream, U_ENC, has_next);
        const Split3 none[1] = {};

        f32x4 acc[2][NT];
        Split3 in[2][KB];
        const char* gw = args.wstream;
        auto encode_xyz = [&](Split3 (&e)[2][KBX]) __attribute__((always_inline)) {
            b3_both([&](auto ct) __attribute__((always_inline)) {
#pragma unroll
                for (int m = 0; m < KBX; ++m) b3_encode_block<FX>(p[ct], lds_bands, m, opaque(g), e[ct][m]);
            });
        };
        // ---- layer1 (no activation)
        b3_both([&](auto ct) __attribute__((always_inline)) { load_bias<NT>(acc[ct], lds_bias, g); });
        {
            Split3 encx[2][KBX];
            encode_xyz(encx);
            gemm_stage_b3w<NT, KBX, 0, NW>(acc[0], acc[1], encx[0], encx[1], none, none, gw, next_of(gw + U_ENC * B3_UNIT, U_HID, true),
                                           lds, slot, carry, wave, lane);
        }
        gw += U_ENC * B3_UNIT;
        b3_both([&](auto ct) __attribute__((always_inline)) { b3_convert<NT, false>(acc[ct], in[ct], lds_walpha + g * (H / 4), false); });

        float sigma[2] = {0.0f, 0.0f};
        const int trunk_iters = density_only ? num_layers - 1 : num_layers;
#pragma unroll 1
        for (int i = 0; i < trunk_iters; ++i) {
            const bool is_feat = i == num_layers - 1;
            const bool skip = !is_feat && ((args.skip_mask >> i) & 1u);
            const bool last_density = density_only && i == num_layers - 2;
            b3_both([&](auto ct) __attribute__((always_inline)) { load_bias<NT>(acc[ct], lds_bias + H * (1 + i), g); });
            {
                const char* after = gw + U_HID * B3_UNIT;
                B3Next nx = next_of(after, U_HID, true);
                if (skip) nx = next_of(after, U_ENC, true);
                else if (is_feat) nx = next_of(after, U_DIR, true);
                else if (last_density) nx = wrap;
                gemm_stage_b3w<NT, KB, 0, NW>(acc[0], acc[1], in[0], in[1], none, none, gw, nx, lds, slot, carry, wave, lane);
                gw = after;
            }
            if (skip) {
                const char* after = gw + U_ENC * B3_UNIT;
                const B3Next nx = last_density ? wrap : next_of(after, U_HID, true);
                Split3 encx[2][KBX];
                encode_xyz(encx);
                gemm_stage_b3w<NT, KBX, 0, NW>(acc[0], acc[1], encx[0], encx[1], none, none, gw, nx, lds, slot, carry, wave, lane);
                gw = after;
            }
            const bool with_alpha = i == num_layers - 2;
            b3_both([&](auto ct) __attribute__((always_inline)) {
                const float part = b3_convert<NT, true>(acc[ct], in[ct], lds_walpha + g * (H / 4), with_alpha);
                if (with_alpha) sigma[ct] = group_sum(part) + tail_bias[0];
            });
        }

        if (density_only) {
            b3_both([&](auto ct) __attribute__((always_inline)) {
                if (valid[ct] && g == 0) args.out[sample[ct]] = sigma[ct];
            });
            continue;
        }

        // ---- layers_dir[0]: cat(feat, dir_enc) -> H/2, relu; fc_rgb + sigmoid on the VALU from the fp32 accumulators
        f32x4 accd[2][NTD];
        Split3 encd[2][KBD];
        b3_both([&](auto ct) __attribute__((always_inline)) {
            load_bias<NTD>(accd[ct], lds_bias + H * (1 + num_layers), g);
            b3_encode_block<FD>(d[ct], lds_bands + 16, 0, opaque(g), encd[ct][0]);
        });
        gemm_stage_b3w<NTD, KB, KBD, NW>(accd[0], accd[1], in[0], in[1], encd[0], encd[1], gw, wrap, lds, slot, carry, wave, lane);
        b3_both([&](auto ct) __attribute__((always_inline)) {
            float rgb[3];
            b3_rgb_head<H, NTD>(accd[ct], lds_wrgb, tail_bias, g, rgb);
            if (valid[ct] && g == 0) {
                f32x4 o4 = {rgb[0], rgb[1], rgb[2], sigma[ct]};
                *reinterpret_cast<f32x4*>(args.out + 4 * sample[ct]) = o4;
            }
        });
    }
}


}  // namespace nm
