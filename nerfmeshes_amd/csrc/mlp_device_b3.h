// OPT-IN precision mode "bf16x3" of the fused MLP: every fp32 product a*b of the GEMMs is replaced by
// six bf16 MFMA products of a three-way split of BOTH operands,
//     x = x1 + x2 + x3,  x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2)      (24 mantissa bits in 3 x 8)
//     a*b ~= a1 b1 + (a1 b2 + a2 b1) + (a1 b3 + a2 b2 + a3 b1)                         (dropped terms <= 2^-24 |a b|)
// accumulated in fp32 by v_mfma_f32_16x16x32_bf16 (16x the fp32 MFMA rate per instruction-flop: 6 of them cost 6/16 of
// the fp32 time).  Error class of fp32 (each product is exact in fp32, the dropped terms are below one fp32 ulp of the
// product), but NOT the reference's arithmetic: the accumulation order inside an MFMA and the dropped cross terms
// differ from an fp32 FMA chain, so results are close to, not bit-identical with, the default path.  Default stays
// fp32 (DESIGN.md section 5 has the parity budget measured for this mode).
//
// Dataflow: the same register-resident chain as the fp32 kernels.  For v_mfma_f32_16x16x32_bf16 lane l holds
//   A: W[row = l & 15][k = 8 (l >> 4) + j], j = 0..7 (8 bf16 = one ds_read_b128)   B: act[k = 8 (l >> 4) + j][sample = l & 15]
//   D: out[row = 4 (l >> 4) + r][sample = l & 15]
// so the 8 k-values lane group g feeds to k-block m are the D registers of tiles 2m and 2m+1 (features 16(2m + j/4) +
// 4g + j%4): bias + ReLU + split happen per lane, nothing crosses lanes; the packer permutes the weight columns.
// Weight stream: per (k-block, output tile) "unit" three 1 KiB planes (W1, W2, W3 images of one ds_read_b128 each);
// chunks of 16 units (48 KiB) through the 3-slot ring of mlp_device_r3.h, scalar-addressed DMA two chunks ahead.
#pragma once
#include "mlp_device_r3.h"

namespace nm {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct Split3 { u32x4 p[3]; };   // one k-block of B operands: 8 values per lane, three bf16 planes

__device__ __forceinline__ unsigned b3_pack(float x, float y) {
    const f32x2 v = {x, y};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));   // v_cvt_pk_bf16_f32 (round to nearest even)
}

// three-way split of a pair of floats into packed bf16 pairs (the subtractions are exact in fp32)
__device__ __forceinline__ void b3_split_pair(float x, float y, unsigned& p1, unsigned& p2, unsigned& p3) {
    p1 = b3_pack(x, y);
    const float rx = x - __uint_as_float(p1 << 16), ry = y - __uint_as_float(p1 & 0xffff0000u);
    p2 = b3_pack(rx, ry);
    const float sx = rx - __uint_as_float(p2 << 16), sy = ry - __uint_as_float(p2 & 0xffff0000u);
    p3 = b3_pack(sx, sy);
}

__device__ __forceinline__ void b3_split8(const float (&v)[8], Split3& out) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        unsigned a, b, c;
        b3_split_pair(v[2 * t], v[2 * t + 1], a, b, c);
        out.p[0][t] = a; out.p[1][t] = b; out.p[2][t] = c;
    }
}

constexpr int B3_UNIT = 3072;          // bytes of one (k-block, tile) unit: 3 planes x 64 lanes x 16 B
constexpr int B3_CHUNK_UNITS = 16;
constexpr int B3_SLOT = B3_CHUNK_UNITS * B3_UNIT;   // 48 KiB

struct B3Next { const char* s0; int b0; const char* s1; int b1; };

// One stage: acc[NT] += W_stage * B, B = KB1 k-blocks of b1 followed by KB2 of b2.  Units are streamed k-block-major
// (all NT tiles of k-block 0, then k-block 1, ...); `carry` holds the A planes of the next two units across chunk and
// stage boundaries (same scheme as gemm_stage3).
// CU: units per chunk (B3_CHUNK_UNITS = 16 for the 256-wide networks; 8 / 3 for the 128- / 64-wide ones, whose stages would
// not span two chunks of 16 units: round 5)
template <int NT, int KB1, int KB2, int NW, int CU = B3_CHUNK_UNITS>
__device__ __forceinline__ void gemm_stage_b3(f32x4 (&acc)[NT], const Split3 (&b1)[KB1], const Split3 (&b2)[(KB2 > 0 ? KB2 : 1)],
                                              const char* gw, const B3Next nx, char* lds, int& slot, u32x4 (&carry)[2][3],
                                              int wave, int lane) {
    constexpr int B3_CHUNK_UNITS = CU;                 // shadows the namespace constants inside this function
    constexpr int B3_SLOT = CU * B3_UNIT;
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
            const Split3& b = m < KB1 ? b1[m < KB1 ? m : 0] : b2[m >= KB1 ? m - KB1 : 0];
            const bf16x8 a1 = __builtin_bit_cast(bf16x8, ab[r0][0]), a2 = __builtin_bit_cast(bf16x8, ab[r0][1]),
                         a3 = __builtin_bit_cast(bf16x8, ab[r0][2]);
            const bf16x8 x1 = __builtin_bit_cast(bf16x8, b.p[0]), x2 = __builtin_bit_cast(bf16x8, b.p[1]),
                         x3 = __builtin_bit_cast(bf16x8, b.p[2]);
            f32x4 d = acc[nt];
            d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3, x1, d, 0, 0, 0);   // smallest terms first
            d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, x2, d, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, x3, d, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, x1, d, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, x2, d, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, x1, d, 0, 0, 0);
            acc[nt] = d;
        }
        __syncthreads();
        slot = slot1;
    }
#pragma unroll
    for (int p = 0; p < 3; ++p) { carry[0][p] = ab[UNITS % 3][p]; carry[1][p] = ab[(UNITS + 1) % 3][p]; }
}

// encoding slots of lane group g in k-block m: pairs (sin a, cos a) for a = 16 m + 4 g + t, t = 0..3, a < 3F; then the
// identity coordinates at slots 6F .. 6F+2; zero padding behind them (the packer uses the same map)
template <int F>
__device__ __forceinline__ void b3_encode_block(const float (&x)[3], const float* bands, int m, int g, Split3& out) {
    float v[8];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int a = 16 * m + 4 * g + t;
        if (a < 3 * F) {
            sincosf(x[a / F] * bands[a % F], &v[2 * t], &v[2 * t + 1]);
        } else {
            const int i0 = 2 * a - 6 * F, i1 = i0 + 1;
            v[2 * t] = i0 == 0 ? x[0] : (i0 == 1 ? x[1] : (i0 == 2 ? x[2] : 0.0f));
            v[2 * t + 1] = i1 == 0 ? x[0] : (i1 == 1 ? x[1] : (i1 == 2 ? x[2] : 0.0f));
        }
    }
    b3_split8(v, out);
}

// accumulators (bias included) -> [ReLU] -> split B operands of the next layer; optionally the fc_alpha GEMV on the
// fp32 values on the way (returns its per-lane partial sum, to be folded over the lane groups)
template <int NT, bool RELU>
__device__ __forceinline__ float b3_convert(const f32x4 (&acc)[NT], Split3 (&out)[NT / 2], const float* walpha_g, bool with_alpha) {
    float part = 0.0f;
#pragma unroll
    for (int m = 0; m < NT / 2; ++m) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float a = acc[2 * m + j / 4][j % 4];
            v[j] = RELU ? fmaxf(a, 0.0f) : a;
        }
        if (with_alpha) {
#pragma unroll
            for (int j = 0; j < 8; ++j) part = fmaf(v[j], walpha_g[4 * (2 * m + j / 4) + j % 4], part);
        }
        b3_split8(v, out[m]);
    }
    return part;
}

template <int H, int FX, int FD, int NW, int CU = B3_CHUNK_UNITS>
__global__ __launch_bounds__(NW * 64, 2) void mlp_kernel_b3(const MlpArgs args, const int num_layers, const int density_only) {
    constexpr int B3_CHUNK_UNITS = CU;
    constexpr int B3_SLOT = CU * B3_UNIT;
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
    // the frequency bands as an LDS table: read where they are used (held in registers across the sample loop they were
    // what the compiler spilled to scratch)
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
    const int64_t wg_iters = (args.n + NW * 16 - 1) / (NW * 16);
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
        const int64_t sample = (it * NW + wave) * 16 + col;
        const bool valid = sample < args.n;
        const SamplePD smp = fetch_sample(args, valid ? sample : args.n - 1);
        const float p[3] = {smp.px, smp.py, smp.pz}, d[3] = {smp.dx, smp.dy, smp.dz};
        // the lane group as a value the compiler cannot see through at each encoding site: the per-lane slot arithmetic
        // (a / F, a % F, a < 3F) is then computed where it is used instead of being hoisted out of the sample and trunk
        // loops and spilled (with the band table in LDS and the skip layer's re-encoding: 246 VGPRs, no scratch; was 104 B)
        auto opaque = [](int v) { asm volatile("" : "+v"(v)); return v; };
        const B3Next wrap = next_of(args.wstream, U_ENC, has_next);
        const Split3 none[1] = {};

        f32x4 acc[NT];
        Split3 in[KB];
        const char* gw = args.wstream;
        // ---- layer1 (no activation)
        load_bias<NT>(acc, lds_bias, g);
        {
            Split3 encx[KBX];
#pragma unroll
            for (int m = 0; m < KBX; ++m) b3_encode_block<FX>(p, lds_bands, m, opaque(g), encx[m]);
            gemm_stage_b3<NT, KBX, 0, NW, CU>(acc, encx, none, gw, next_of(gw + U_ENC * B3_UNIT, U_HID, true), lds, slot, carry, wave, lane);
        }
        gw += U_ENC * B3_UNIT;
        b3_convert<NT, false>(acc, in, lds_walpha + g * (H / 4), false);

        float sigma = 0.0f;
        const int trunk_iters = density_only ? num_layers - 1 : num_layers;
#pragma unroll 1
        for (int i = 0; i < trunk_iters; ++i) {
            const bool is_feat = i == num_layers - 1;
            const bool skip = !is_feat && ((args.skip_mask >> i) & 1u);
            const bool last_density = density_only && i == num_layers - 2;
            load_bias<NT>(acc, lds_bias + H * (1 + i), g);
            {
                const char* after = gw + U_HID * B3_UNIT;
                B3Next nx = next_of(after, U_HID, true);
                if (skip) nx = next_of(after, U_ENC, true);
                else if (is_feat) nx = next_of(after, U_DIR, true);
                else if (last_density) nx = wrap;
                gemm_stage_b3<NT, KB, 0, NW, CU>(acc, in, none, gw, nx, lds, slot, carry, wave, lane);
                gw = after;
            }
            if (skip) {
                const char* after = gw + U_ENC * B3_UNIT;
                const B3Next nx = last_density ? wrap : next_of(after, U_HID, true);
                // the encoding is recomputed here instead of being held across the trunk (24 registers for 8 sincosf);
                // same values, same bits
                Split3 encx[KBX];
#pragma unroll
                for (int m = 0; m < KBX; ++m) b3_encode_block<FX>(p, lds_bands, m, opaque(g), encx[m]);
                gemm_stage_b3<NT, KBX, 0, NW, CU>(acc, encx, none, gw, nx, lds, slot, carry, wave, lane);
                gw = after;
            }
            // fc_alpha reads the output of layers_xyz[L-2] (models.py:71): its GEMV rides on this conversion
            const bool with_alpha = i == num_layers - 2;
            const float part = b3_convert<NT, true>(acc, in, lds_walpha + g * (H / 4), with_alpha);
            if (with_alpha) sigma = group_sum(part) + tail_bias[0];
        }

        if (density_only) {
            if (valid && g == 0) args.out[sample] = sigma;
            continue;
        }

        // ---- layers_dir[0]: cat(feat, dir_enc) -> H/2, relu; fc_rgb + sigmoid on the VALU from the fp32 accumulators
        f32x4 accd[NTD];
        load_bias<NTD>(accd, lds_bias + H * (1 + num_layers), g);
        Split3 encd[KBD];
        b3_encode_block<FD>(d, lds_bands + 16, 0, opaque(g), encd[0]);
        gemm_stage_b3<NTD, KB, KBD, NW, CU>(accd, in, encd, gw, wrap, lds, slot, carry, wave, lane);
        float v[4 * NTD];
        acc_to_operand<NTD, true>(accd, v);
        float rgb[3];
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
        if (valid && g == 0) {
            f32x4 o4 = {rgb[0], rgb[1], rgb[2], sigma};
            *reinterpret_cast<f32x4*>(args.out + 4 * sample) = o4;
        }
    }
}

}  // namespace nm
