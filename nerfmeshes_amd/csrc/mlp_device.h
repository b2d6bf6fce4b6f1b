// Device-side building blocks of the fused MLP kernels (forward: nerf_mlp.hip, training: nerf_train.hip):
// MFMA operand layouts, the L2 -> LDS weight-stream ring, GEMM stages, positional encoding.
// See the header comment of nerf_mlp.hip for the dataflow.
#pragma once
#include "nm_internal.h"

namespace nm {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int H_, int FX_, int FD_, int KCH_ = KC>
struct Net {
    static constexpr int H = H_, FX = FX_, FD = FD_, KCH = KCH_;
    static constexpr int NT = H / 16;                // 16-row output tiles of a hidden layer
    static constexpr int KH = H / 4;                 // k-steps across a hidden activation
    static constexpr int EX = (3 * FX + 1) / 2 + 1;  // k-steps across the xyz encoding
    static constexpr int ED = (3 * FD + 1) / 2 + 1;  // k-steps across the dir encoding
    static constexpr int NTD = H / 32;               // tiles of the H/2-wide view layer
    static constexpr int KD = H / 8;                 // k-steps across the view layer output
    static constexpr int STEP = NT * 256;            // bytes of A operands per k-step (hidden out)
    static constexpr int STEPD = NTD * 256;
    static constexpr int LDSBUF = KCH * STEP;        // one ring slot
    static constexpr int L1_FIRST = (EX < KCH ? EX : KCH) * STEP;
    static constexpr int DIR_FIRST = (KH + ED < KCH ? KH + ED : KCH) * STEPD;
    static_assert(KH >= KCH, "a hidden layer must span at least one full chunk");
};

// ---- weight stream: HBM/L2 -> LDS DMA, 1 KiB per wave-instruction, LDS image == stream image ------
// The DMA is `buffer_load_dwordx4 off, s[rsrc], s_offset lds` through a descriptor with ADD_TID_ENABLE and stride 16:
// the hardware adds lane * 16 to the address, so the instruction reads NO address VGPR.  That matters more than
// anything else about the stream: a VMEM instruction that fetches per-lane 64-bit addresses from the register file
// (global_load_lds_dwordx4 v[a:a+1], off) holds up the MFMA issue of its SIMD for ~60 cycles, whatever it moves and
// whichever wave issues it -- 2380 pieces per 128-sample tile = 6 % of the kernel (profiles/r02_mlp_variants.json);
// the scalar-addressed form costs 1.3 %.  (In this mode the descriptor's DATA_FORMAT bits are stride[17:14]: left 0.)
template <int NW>
__device__ __forceinline__ void stream_to_lds(const char* src, char* dst, int bytes, int wave, int lane) {
    (void)lane;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), (short)16, 0x7fffffff, 1 << 23);
    const int units = (bytes + 1023) >> 10;
    for (int u = wave; u < units; u += NW)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(dst + u * 1024), 16, 0,
                                                 u * 1024, 0, 0);
}

// round-1 form (per-lane 64-bit addresses in VGPRs), kept for the A/B variants of the ablation library
template <int NW>
__device__ __forceinline__ void stream_to_lds_vaddr(const char* src, char* dst, int bytes, int wave, int lane) {
    const int units = (bytes + 1023) >> 10;
    for (int u = wave; u < units; u += NW) {
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(src + (size_t)u * 1024 + lane * 16),
            (__attribute__((address_space(3))) void*)(dst + u * 1024), 16, 0, 0);
    }
}

// One GEMM "stage": acc[NT tiles] += W_stage * B, B = KS1 registers of b1 followed by KS2 of b2.
// On entry chunk 0 of the stage is resident in ring slot `par`; on exit the chunk described by
// (tail_src, tail_bytes) -- the first chunk of whatever runs next -- is resident in slot `par`.
// Per chunk: start the DMA of the following chunk, run this chunk's MFMAs with the A operands of the NEXT
// k-step already in flight (PIPE; pinned with sched_barrier -- hipcc otherwise sinks the ds_reads next to
// their use, and the two waves of a SIMD, released together by the barrier, then expose the LDS latency
// together every 8 MFMAs), one barrier.  (The cursor is SGPR arithmetic on purpose: a chunk table fetched
// with s_load costs 3 % -- its s_waitcnt lgkmcnt(0) also drains the in-flight ds_reads.)
//
// STORE (training): b1 -- the previous layer's activation / delta, live in registers for this whole stage anyway --
// is written to `store_row` (this lane's row-major position, tape layout) a few tiles per chunk instead of as one
// 16-instruction burst before the stage: the bursts of the 8 waves of a workgroup (128 KB) outrun the HBM write
// bandwidth and the next barrier's s_waitcnt vmcnt(0) -- gfx9 counts stores and loads together -- waits for them
// (10 % of the delta kernel); spread out, each store has a chunk's worth of MFMAs (1.7 us) to retire.
template <int NT, int KS1, int KS2, int NW, int LDSBUF, int KCH, bool PIPE, bool SPREAD = false, int ABL = 0,
          bool STORE = false>
__device__ __forceinline__ void gemm_stage(f32x4 (&acc)[NT], const float (&b1)[KS1],
                                           const float (&b2)[(KS2 > 0 ? KS2 : 1)], const char* gw,
                                           const char* tail_src, int tail_bytes, char* lds, int& par,
                                           int wave, int lane, float* store_row = nullptr) {
    constexpr int KS = KS1 + KS2;
    constexpr int NCH = (KS + KCH - 1) / KCH;
    constexpr int VW = NT >= 4 ? 4 : NT;  // A operands fetched per LDS read
    constexpr int NB = NT / VW;
    constexpr int STEP_BYTES = NT * 256;
    typedef float avec __attribute__((ext_vector_type(VW)));
    static_assert(NT % VW == 0 && (VW == 4 || VW == 2), "tile count");
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int steps = (KS - c * KCH) < KCH ? (KS - c * KCH) : KCH;
        char* next_slot = lds + (par ^ 1) * LDSBUF;
        const char* next_src;
        int next_bytes;
        if (c + 1 < NCH) {
            const int nsteps = (KS - (c + 1) * KCH) < KCH ? (KS - (c + 1) * KCH) : KCH;
            next_src = gw + (c + 1) * KCH * STEP_BYTES; next_bytes = nsteps * STEP_BYTES;
        } else {
            next_src = tail_src; next_bytes = tail_bytes;
        }
        // DMA of the following chunk: either all of this wave's 1 KiB pieces up front, or (SPREAD) one piece per
        // k-step inside the MFMA stream -- issued together right after the barrier they keep BOTH waves of a
        // SIMD away from the matrix pipe for ~100 cycles per piece, 73 times per tile
        const int next_units = (next_bytes + 1023) >> 10;
        int next_u = wave;
        if constexpr (!SPREAD && !(ABL & 4)) {
            if constexpr (ABL & 8) stream_to_lds_vaddr<NW>(next_src, next_slot, next_bytes, wave, lane);
            else stream_to_lds<NW>(next_src, next_slot, next_bytes, wave, lane);
        }
        if constexpr (STORE) {
            constexpr int TILES = KS1 / 4, PER_CHUNK = (TILES + NCH - 1) / NCH;
            if (store_row) {
#pragma unroll
                for (int q = 0; q < PER_CHUNK; ++q) {
                    const int nt = c * PER_CHUNK + q;
                    if (nt < TILES) {
                        const f32x4 v4 = {b1[4 * nt], b1[4 * nt + 1], b1[4 * nt + 2], b1[4 * nt + 3]};
                        *reinterpret_cast<f32x4*>(store_row + 16 * nt) = v4;
                    }
                }
            }
        }
        const char* buf = lds + par * LDSBUF + lane * (VW * 4);
        avec a_next[NB];
        if constexpr (PIPE) {
#pragma unroll
            for (int blk = 0; blk < NB; ++blk) a_next[blk] = *reinterpret_cast<const avec*>(buf + blk * (64 * VW * 4));
        }
#pragma unroll
        for (int ks = 0; ks < steps; ++ks) {
            const int s = c * KCH + ks;
            const float b = s < KS1 ? b1[s < KS1 ? s : 0] : b2[s >= KS1 ? s - KS1 : 0];
            avec a_cur[NB];
#pragma unroll
            for (int blk = 0; blk < NB; ++blk) {
                if constexpr (PIPE) a_cur[blk] = a_next[blk];
                else a_cur[blk] = *reinterpret_cast<const avec*>(buf + (ks * NB + blk) * (64 * VW * 4));
            }
            if constexpr (PIPE) {
                if (ks + 1 < steps) {
#pragma unroll
                    for (int blk = 0; blk < NB; ++blk)
                        a_next[blk] = *reinterpret_cast<const avec*>(buf + ((ks + 1) * NB + blk) * (64 * VW * 4));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (SPREAD) {
                if (next_u < next_units) {
                    __builtin_amdgcn_global_load_lds(
                        (const __attribute__((address_space(1))) void*)(next_src + (size_t)next_u * 1024 + lane * 16),
                        (__attribute__((address_space(3))) void*)(next_slot + next_u * 1024), 16, 0, 0);
                    next_u += NW;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int blk = 0; blk < NB; ++blk)
#pragma unroll
                for (int q = 0; q < VW; ++q)
                    acc[blk * VW + q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[blk][q], b, acc[blk * VW + q], 0, 0, 0);
        }
        if constexpr (SPREAD) {   // pieces beyond the chunk's k-step count (short chunks)
            for (; next_u < next_units; next_u += NW)
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(next_src + (size_t)next_u * 1024 + lane * 16),
                    (__attribute__((address_space(3))) void*)(next_slot + next_u * 1024), 16, 0, 0);
        }
        if constexpr (!(ABL & 2)) __syncthreads();  // drains the DMA (vmcnt(0)) and releases slot `par` for the next fill
        par ^= 1;
    }
}

template <int NT>
__device__ __forceinline__ void load_bias(f32x4 (&acc)[NT], const float* bias, int g) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = *reinterpret_cast<const f32x4*>(bias + 16 * nt + 4 * g);
}

template <int NT, bool RELU>
__device__ __forceinline__ void acc_to_operand(const f32x4 (&acc)[NT], float (&op)[4 * NT]) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) op[4 * nt + r] = RELU ? fmaxf(acc[nt][r], 0.0f) : acc[nt][r];
}

// Positional encoding, laid out as MFMA B operands.  k-step s < STEPS-1 carries two encoding
// arguments a0 = 2s, a1 = 2s+1 (a = coord * F + freq, the reference's coordinate-major order):
// lane group 0: sin(a0)  1: cos(a0)  2: sin(a1)  3: cos(a1).  The last step carries (x, y, z, 0).
// sincosf yields both halves, and a sin group and its cos group (lanes l, l ^ 16) want the two halves of the SAME
// arguments: over a pair of k-steps (s, s + 1) the sin group evaluates the argument of step s, the cos group that of
// step s + 1, each keeps the half it needs and hands the other across with one ds_swizzle -- 11 sincosf per lane for the
// 8x256 network instead of 21, the same function on the same products, so the results are bit for bit what they were.
// (Round 3: VALU issue time adds to matrix time -- DESIGN.md 3.1 -- and 58 % of a narrow network's VALU work was this.)
template <int F, int STEPS, int ABL = 0>
__device__ __forceinline__ void encode(float (&enc)[STEPS], const float (&x)[3], const float* bands, int g) {
    const bool hi = (g >> 1) != 0;
    const bool want_cos = (g & 1) != 0;
    constexpr int NS = STEPS - 1;
    auto argument = [&](int s) -> float {      // what this lane's pair of groups encodes at k-step s (s: compile-time)
        const int a0 = 2 * s, a1 = 2 * s + 1;
        const float x0 = x[a0 / F] * bands[a0 % F];
        const float x1 = (a1 < 3 * F) ? x[(a1 < 3 * F ? a1 : 0) / F] * bands[(a1 < 3 * F ? a1 : 0) % F] : 0.0f;
        return hi ? x1 : x0;
    };
#pragma unroll
    for (int s = 0; s + 1 < NS; s += 2) {
        const float mine = want_cos ? argument(s + 1) : argument(s);
        float sv, cv;
        if constexpr (ABL & 1) { sv = mine; cv = mine + 1.0f; }
        else sincosf(mine, &sv, &cv);
        const float give = want_cos ? sv : cv;
        const float got = __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(give), 0x401f));   // lane ^ 16
        enc[s] = want_cos ? got : sv;
        enc[s + 1] = want_cos ? cv : got;
    }
    if constexpr (NS % 2 == 1) {
        float sv, cv;
        const float mine = argument(NS - 1);
        if constexpr (ABL & 1) { sv = mine; cv = mine + 1.0f; }
        else sincosf(mine, &sv, &cv);
        enc[NS - 1] = want_cos ? cv : sv;
    }
    enc[STEPS - 1] = g == 0 ? x[0] : (g == 1 ? x[1] : (g == 2 ? x[2] : 0.0f));
}

// TAPE: the encoding a lane holds as B operands (encode() above), written as this sample's row in the reference's column order
// [x y z | sin, argument-major | cos] (modules.py:26-34): k-step s < STEPS - 1 carries sin / cos of arguments 2s (lane groups
// 0 / 1) and 2s + 1 (groups 2 / 3), the last step the raw coordinates.  One dword per lane and k-step; the four groups of a
// sample fill two 8-byte runs of its row.  The weight-gradient kernels contract the layer1 / skip / view deltas with these rows;
// until round 6 a separate pass recomputed them (nm_encode_samples_strided: 0.15 ms of a 2048-ray iteration).
template <int F, int STEPS>
__device__ __forceinline__ void store_encoding_row(float* row, const float (&enc)[STEPS], int g) {
    // Two k-steps per 8-byte store: the arguments 2s and 2s + 1 of a step sit in lanes l and l + 32 (groups 0 / 2 for the sines,
    // 1 / 3 for the cosines) and are neighbours in the row, so ONE v_permlane32_swap per step hands each half of the wave the
    // other's value; the lower half then stores step s, the upper half step s + 1 -- 13 store instructions per sample tile of the
    // 8x256 network instead of 23 dword ones (each VGPR-addressed store holds up the SIMD's MFMA issue, section 3.5).
    const bool upper = g >= 2;
    float* sincos = row ? row + 3 + ((g & 1) ? 3 * F : 0) : nullptr;
    constexpr int NS = STEPS - 1;
#pragma unroll
    for (int s = 0; s < NS; s += 2) {
        const auto lo = __builtin_amdgcn_permlane32_swap(__float_as_uint(enc[s]), __float_as_uint(enc[s]), false, false);
        float first = enc[s], second = __uint_as_float(lo[1]);              // lower half: (own, partner) of step s
        int a = 2 * s;
        if (s + 1 < NS) {
            const auto hi = __builtin_amdgcn_permlane32_swap(__float_as_uint(enc[s + 1]), __float_as_uint(enc[s + 1]), false, false);
            if (upper) { first = __uint_as_float(hi[0]); second = enc[s + 1]; a = 2 * s + 2; }   // upper half: (partner, own) of step s + 1
        } else if (upper) {
            a = 3 * F;                                                       // no step s + 1: nothing to store
        }
        if (row && a + 1 < 3 * F) {
            typedef float f32x2 __attribute__((ext_vector_type(2), aligned(4)));
            *reinterpret_cast<f32x2*>(sincos + a) = f32x2{first, second};
        } else if (row && a < 3 * F) {
            sincos[a] = first;
        }
    }
    if (row && g < 3) row[g] = enc[STEPS - 1];
}

__device__ __forceinline__ float group_sum(float v) {  // sum over the 4 lane groups (lanes l, l^16, l^32, l^48)
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}

template <int H>
__device__ __forceinline__ float alpha_gemv(const float (&in)[H / 4], const float* walpha, int g) {
    float part = 0.0f;   // fc_alpha (models.py:71): 1-row GEMV on the VALU + lane-group reduction
    const float* wa = walpha + g * (H / 4);
#pragma unroll
    for (int s = 0; s < H / 4; s += 4) {
        const f32x4 w4 = *reinterpret_cast<const f32x4*>(wa + s);
#pragma unroll
        for (int q = 0; q < 4; ++q) part = fmaf(in[s + q], w4[q], part);
    }
    return group_sum(part);
}

// A network without view directions (models.py:77-79): the colour rows of fc_out over the trunk output, three more GEMVs
// in fc_alpha's operand layout, sigmoid, and the (rgb, sigma) row -- `density_only` == 2 in the kernels below.
template <int H>
__device__ __forceinline__ void flat_head(const MlpArgs& args, const float (&in)[H / 4], const float* wrows,
                                          const float* tail_bias, float sigma, int64_t sample, bool valid, int g) {
    float rgb[3];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        const float x = alpha_gemv<H>(in, wrows + ch * H, g) + tail_bias[1 + ch];
        rgb[ch] = 1.0f / (1.0f + expf(-x));
    }
    if (valid && g == 0) {
        f32x4 o4 = {rgb[0], rgb[1], rgb[2], sigma};
        *reinterpret_cast<f32x4*>(args.out + 4 * sample) = o4;
    }
}

// ---- training tape helpers ---------------------------------------------------------------------------------
// D-layout registers (tile nt, register r = feature 16*nt + 4*g + r of this lane's sample) <-> row-major [sample][width]
template <int NT>
__device__ __forceinline__ void store_rows(float* base, int width, int64_t sample, bool valid, const float (&op)[4 * NT],
                                           int g) {
    if (!valid) return;
    float* row = base + sample * width + 4 * g;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const f32x4 v = {op[4 * nt], op[4 * nt + 1], op[4 * nt + 2], op[4 * nt + 3]};
        *reinterpret_cast<f32x4*>(row + 16 * nt) = v;
    }
}

template <int COUNT>
__device__ __forceinline__ uint64_t positive_mask(const float (&op)[COUNT]) {
    uint64_t m = 0;
#pragma unroll
    for (int j = 0; j < COUNT; ++j) m |= op[j] > 0.0f ? (uint64_t(1) << j) : uint64_t(0);
    return m;
}

// The sample a lane evaluates: position p and view direction d, by input mode.  Returned BY VALUE as six scalars: with
// `float (&p)[3], (&d)[3]` out-parameters filled inside the mode branches, hipcc kept p[2] and d[2] in scratch memory
// (an alloca it could not promote) -- 8 B per lane and tile stored through to HBM, three times the kernel's useful
// output in WRITE_SIZE.
struct SamplePD { float px, py, pz, dx, dy, dz; };

__device__ __forceinline__ SamplePD fetch_sample(const MlpArgs& args, int64_t sidx) {
    SamplePD s;
    if (args.mode == MODE_POINTS) {
        s.px = args.a[3 * sidx]; s.py = args.a[3 * sidx + 1]; s.pz = args.a[3 * sidx + 2];
        s.dx = args.b[3 * sidx]; s.dy = args.b[3 * sidx + 1]; s.dz = args.b[3 * sidx + 2];
    } else if (args.mode == MODE_RAYS) {
        const int64_t ray = sidx / args.samples;
        const float t = args.c[sidx];
        const float* o = args.a + (args.origins_per_ray ? 3 * ray : 0);
        s.dx = args.b[3 * ray]; s.dy = args.b[3 * ray + 1]; s.dz = args.b[3 * ray + 2];
        // -ffp-contract=off: two roundings, as torch (model_helpers.py:33)
        const float tx = s.dx * t, ty = s.dy * t, tz = s.dz * t;
        s.px = o[0] + tx; s.py = o[1] + ty; s.pz = o[2] + tz;
    } else if (args.mode == MODE_VIEW) {
        const int64_t ray = sidx / args.samples;
        const float t = args.c[sidx];
        float o[3], d[3];
        nm_gen_ray(args.gen, ray, o, d);
        s.dx = d[0]; s.dy = d[1]; s.dz = d[2];
        const float tx = d[0] * t, ty = d[1] * t, tz = d[2] * t;
        s.px = o[0] + tx; s.py = o[1] + ty; s.pz = o[2] + tz;
    } else {
        const int64_t flat = args.first + sidx;
        const int64_t plane = (int64_t)args.n1 * args.n2;
        const int64_t i0 = flat / plane;
        const int64_t rem = flat - i0 * plane;
        const int i1 = (int)(rem / args.n2), i2 = (int)(rem - (int64_t)i1 * args.n2);
        s.px = args.a[i0]; s.py = args.b[i1]; s.pz = args.c[i2];
        s.dx = s.px; s.dy = s.py; s.dz = s.pz;   // mesh_nerf.py:45: sample_points(samples, samples)
    }
    return s;
}

// ---- the fused forward kernel (TAPE: also records the activations the backward pass needs) ---------------
// FLAT: the instantiation that also serves use_viewdirs = 0 networks (`density_only` == 2, see flat_head); a separate
// instantiation so that the production kernels compile exactly as they did without it
template <int H, int FX, int FD, int NW, int KCH, bool PIPE, bool KEEP_ENC, bool LBIAS, bool SPREAD, int ABL, bool TAPE, bool FLAT = false>
// (occupancy: see mlp_kernel3 -- inference instances up to 128 wide are compiled for four waves per SIMD; the taping ones, whose
// extra registers would all be spilled for it, measured no gain and stay at two)
__global__ __launch_bounds__(NW * 64, (H <= 128 && !TAPE) ? 4 : 2) void mlp_kernel(const MlpArgs args, const int num_layers,
                                                      const int density_only) {
    using N = Net<H, FX, FD, KCH>;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    // LBIAS: every bias of the network lives in LDS behind the ring for the whole launch (a bias fetched from
    // L2 at the top of a layer is ~1 us of exposed latency in front of that layer's first MFMA)
    float* lds_bias = reinterpret_cast<float*>(lds + 2 * N::LDSBUF);
    const int nbias = H * (1 + num_layers) + H / 2 + 4;   // ... | fc_alpha.bias | fc_rgb.bias[3]
    float* lds_walpha = lds_bias + nbias;          // [4][H/4]
    float* lds_wrgb = lds_walpha + H;              // [3][4][H/8]
    if constexpr (LBIAS) {
        for (int i = threadIdx.x; i < nbias; i += NW * 64) lds_bias[i] = args.bias[i];
        for (int i = threadIdx.x; i < H; i += NW * 64) lds_walpha[i] = args.walpha[i];
        for (int i = threadIdx.x; i < ((FLAT && density_only == 2) ? 3 * H : 3 * H / 2); i += NW * 64) lds_wrgb[i] = args.wrgb[i];
    }
    const float* bias_src = LBIAS ? lds_bias : args.bias;
    const float* walpha_src = LBIAS ? lds_walpha : args.walpha;
    const float* wrgb_src = LBIAS ? lds_wrgb : args.wrgb;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, col = lane & 15;
    const float* tail_bias = bias_src + nbias - 4;

    const int64_t wg_iters = (args.n + NW * 16 - 1) / (NW * 16);
    int par = 0;
    const char* gw = args.wstream;
    if ((int64_t)blockIdx.x < wg_iters) stream_to_lds<NW>(gw, lds, N::L1_FIRST, wave, lane);

    for (int64_t it = blockIdx.x; it < wg_iters; it += gridDim.x) {
        const bool has_next = it + gridDim.x < wg_iters;
        const int64_t sample = (it * NW + wave) * 16 + col;
        const bool valid = sample < args.n;
        const int64_t sidx = valid ? sample : args.n - 1;

        // ---- prologue: fetch the sample
        const SamplePD smp = fetch_sample(args, sidx);
        const float p[3] = {smp.px, smp.py, smp.pz}, d[3] = {smp.dx, smp.dy, smp.dz};
        const float dummy[1] = {0.0f};
        float encx_keep[KEEP_ENC ? N::EX : 1];
        if constexpr (KEEP_ENC) encode<FX, N::EX, ABL>(encx_keep, p, args.bands_xyz, g);
        if constexpr (KEEP_ENC && TAPE) store_encoding_row<FX, N::EX>((valid && args.tape_encx) ? args.tape_encx + sample * 64 : nullptr, encx_keep, g);

        f32x4 acc[N::NT];
        float in[N::KH];
        __syncthreads();  // first chunk of layer1 resident (its DMA was issued one tile earlier)

        // ---- layer1: xyz_enc -> H, no activation (models.py:62)
        load_bias<N::NT>(acc, bias_src, g);
        if constexpr (KEEP_ENC) {
            gemm_stage<N::NT, N::EX, 0, NW, N::LDSBUF, KCH, PIPE, SPREAD, ABL>(acc, encx_keep, dummy, gw, gw + N::EX * N::STEP,
                                                                   N::LDSBUF, lds, par, wave, lane);
        } else {   // the encoding registers live only for this stage; the skip layer recomputes them
            float encx[N::EX];
            encode<FX, N::EX, ABL>(encx, p, args.bands_xyz, g);
            if constexpr (TAPE) store_encoding_row<FX, N::EX>((valid && args.tape_encx) ? args.tape_encx + sample * 64 : nullptr, encx, g);
            gemm_stage<N::NT, N::EX, 0, NW, N::LDSBUF, KCH, PIPE, SPREAD, ABL>(acc, encx, dummy, gw, gw + N::EX * N::STEP, N::LDSBUF,
                                                                   lds, par, wave, lane);
        }
        gw += N::EX * N::STEP;
        acc_to_operand<N::NT, false>(acc, in);
        const int64_t tile = it * NW + wave;
        // TAPE: every trunk activation is written while the NEXT stage consumes it (gemm_stage STORE): tape_h[i] during
        // trunk stage i, relu(fc_feat) during the view stage
        float* tape_row = (TAPE && valid) ? args.tape_h + sample * H + 4 * g : nullptr;

        // ---- layers_xyz[0 .. L-2], then (full evaluation only) fc_feat as iteration L-1 (models.py:63-70)
        float sigma = 0.0f;
        const int trunk_iters = density_only ? num_layers - 1 : num_layers;
#pragma unroll 1
        for (int i = 0; i < trunk_iters; ++i) {
            const bool is_feat = i == num_layers - 1;
            if (is_feat) sigma = alpha_gemv<H>(in, walpha_src, g) + tail_bias[0];   // on the pre-feature activation
            const bool skip = !is_feat && ((args.skip_mask >> i) & 1u);
            const bool last_density = density_only && i == num_layers - 2;
            load_bias<N::NT>(acc, bias_src + H * (1 + i), g);
            {
                const char* tsrc = gw + N::KH * N::STEP;
                int tbytes = N::LDSBUF;
                if (skip) tbytes = N::L1_FIRST;               // the skip layer's encoding columns follow
                else if (is_feat) tbytes = N::DIR_FIRST;      // view layer follows
                else if (last_density) { tsrc = args.wstream; tbytes = has_next ? N::L1_FIRST : 0; }
                gemm_stage<N::NT, N::KH, 0, NW, N::LDSBUF, KCH, PIPE, SPREAD, ABL, TAPE>(
                    acc, in, dummy, gw, tsrc, tbytes, lds, par, wave, lane,
                    (tape_row && !(i == 0 && args.tape_skip_h0)) ? tape_row + (int64_t)i * args.n * H : nullptr);
                gw += N::KH * N::STEP;
            }
            if (skip) {  // cat(hidden, xyz_enc): the encoding columns of layers_xyz[i] (models.py:64-65)
                const char* tsrc = gw + N::EX * N::STEP;
                int tbytes = N::LDSBUF;
                if (last_density) { tsrc = args.wstream; tbytes = has_next ? N::L1_FIRST : 0; }
                if constexpr (KEEP_ENC) {
                    gemm_stage<N::NT, N::EX, 0, NW, N::LDSBUF, KCH, PIPE, SPREAD, ABL>(acc, encx_keep, dummy, gw, tsrc, tbytes, lds, par, wave, lane);
                } else {
                    float encx[N::EX];
                    encode<FX, N::EX, ABL>(encx, p, args.bands_xyz, g);
                    gemm_stage<N::NT, N::EX, 0, NW, N::LDSBUF, KCH, PIPE, SPREAD, ABL>(acc, encx, dummy, gw, tsrc, tbytes, lds, par, wave, lane);
                }
                gw += N::EX * N::STEP;
            }
            acc_to_operand<N::NT, true>(acc, in);
            if constexpr (TAPE) {
                if (tile < args.tiles) args.mask_h[((int64_t)i * args.tiles + tile) * 64 + lane] = positive_mask(in);
            }
        }

        if (density_only) {
            sigma = alpha_gemv<H>(in, walpha_src, g) + tail_bias[0];
            if (FLAT && density_only == 2) {   // use_viewdirs = 0
                flat_head<H>(args, in, wrgb_src, tail_bias, sigma, sample, valid, g);
                // TAPE: the trunk's last activation has no stage behind it that would write it while consuming it
                if constexpr (TAPE) store_rows<N::NT>(args.tape_h + (int64_t)(num_layers - 1) * args.n * H, H, sample, valid, in, g);
            } else if (valid && g == 0) args.out[sample] = sigma;
            gw = args.wstream;
            continue;
        }

        // ---- layers_dir[0]: cat(feat, dir_enc) -> H/2, relu (models.py:72-74)
        f32x4 accd[N::NTD];
        float v[N::KD];
        load_bias<N::NTD>(accd, bias_src + H * (1 + num_layers), g);
        float encd[N::ED];
        encode<FD, N::ED, ABL>(encd, d, args.bands_dir, g);
        if constexpr (TAPE) store_encoding_row<FD, N::ED>((valid && args.tape_encd) ? args.tape_encd + sample * 64 : nullptr, encd, g);
        gemm_stage<N::NTD, N::KH, N::ED, NW, N::LDSBUF, KCH, PIPE, SPREAD, ABL, TAPE>(
            accd, in, encd, gw, args.wstream, has_next ? N::L1_FIRST : 0, lds, par, wave, lane,
            (TAPE && valid) ? args.tape_feat + sample * H + 4 * g : nullptr);
        gw = args.wstream;
        acc_to_operand<N::NTD, true>(accd, v);
        if constexpr (TAPE) {
            store_rows<N::NTD>(args.tape_v, args.tape_v_ld, sample, valid, v, g);
            if (tile < args.tiles) args.mask_v[tile * 64 + lane] = positive_mask(v);
        }

        // ---- fc_rgb + sigmoid (models.py:75), 3-row GEMV on the VALU
        float rgb[3];
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            float part = 0.0f;
            const float* wr = wrgb_src + (ch * 4 + g) * N::KD;
#pragma unroll
            for (int s = 0; s < N::KD; s += 4) {
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
