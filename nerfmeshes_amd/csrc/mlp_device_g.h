// Generic-shape instantiation family of the fused MLP (round 4): every FlexibleNeRFModel the reference's constructor accepts
// (/root/reference/src/nerf/models.py:5-58 -- any hidden_size up to 512, 0..15 encoding functions per input, include_input_*
// on or off, with or without view directions) runs on the same register-resident dataflow as the tuned plans of
// nerf_mlp.hip; what is a template parameter there and a RUNTIME value here:
//
//   * hidden_size -> a width class NT (16-row MFMA tiles, 16 * NT >= hidden_size; classes in nerf_mlp_generic*.hip).  The packer
//     pads rows / columns / biases beyond the real width with zeros: relu(0 * x + 0) = 0 feeds zero columns, so the padded
//     network computes exactly the real one.  The view layer (hidden_size // 2 rows) is padded to NTD = ceil(NT / 2) tiles.
//     A-operand blocks are 4 tiles (one ds_read_b128 per lane); a trailing block of NT % 4 tiles skips its unused MFMAs.
//   * num_encoding_fn_* / include_input_* -> encodings live in G_ENC_STEPS = 24 registers per lane (longer ones -- more than 15
//     functions, up to 31 -- in two parts, the second evaluated into the same registers when its chunks come up); how many k-steps are
//     real, which argument (coordinate, frequency band) each lane encodes and whether an identity step follows come from a
//     host-built table (LDS-resident).  An encoding's GEMM columns are their own stage of a runtime number of whole
//     KCH-k-step chunks (zero-padded), accumulated into the same tiles as the hidden columns -- as the tuned kernels do for
//     the skip layer; here also for layers_dir[0] (cat(feat, view): two stages).
//   * use_viewdirs = 0 is the runtime `density_only == 2` branch (flat_head), not a separate instantiation.
//
// The values are the tuned kernels': the same sincosf on the same products, fp32 MFMA chains over the same column order
// (the padded k-steps add exact zeros at the end of a chain), so on a menu shape this kernel reproduces the tuned one bit
// for bit (tests/test_gpu_generic.py).  Dataflow = round 1's (2-slot ring, one barrier per chunk, scalar-addressed DMA) with
// round 2's block-granular operand prefetch inside a chunk.  Classes up to 24 tiles (hidden_size <= 384) run 8-wave
// workgroups, two waves per SIMD, like the tuned kernels: 2 * 4 * NT activation + accumulator registers leave room in 256
// up to NT = 20; classes 21 -- 24 re-encode the position at the skip layer instead of holding it and still spill a few dozen
// registers at the stage boundaries, outside the k-step loops -- measured 0.89 -- 0.91 of the peak against 0.80 with one wave per SIMD (the partner
// wave hides a wave's LDS and barrier latency; round 4, profiles/r04_mlp_shapes.json).  Beyond 24 tiles the two arrays alone
// exceed 256: those classes run on mlp_device_gs.h (a layer's output tiles split over a pair of waves; round 5).  The NW = 4
// instantiations of THIS kernel for them -- one wave per SIMD on the 512-register budget -- remain in the ablation library only.
#pragma once
#include "mlp_device.h"

namespace nm {

constexpr int G_ENC_STEPS = 24;     // k-steps of an encoding that live in registers at a time
constexpr int G_ENC_ARGS = 2 * G_ENC_STEPS;
constexpr int G_LONG_VARIANT = 50;  // MlpPlan::variant of the LONG instantiations
constexpr int G_ENC_PARTS = 2;      // an encoding spans up to G_ENC_PARTS * 24 k-steps: (3 F + 1) / 2 + include_input <= 48 (F <= 31; 32 without the
                                    // input); the second part is evaluated when its stage comes up (enc_stages_g), never held

// Dynamic LDS of the generic kernels beyond the weight ring -- ONE definition for the kernels' carve-up and for every launcher
// (nerf_mlp.hip: inference; nerf_train.hip: taping forward, delta kernel).  Forward / taping: biases (layer1 | layers_xyz.* |
// fc_feat | layers_dir.0 | fc_alpha.b | fc_rgb.b[3]: HP (1 + L) + HPD + 4), fc_alpha's row (HP), the colour rows (3 HP: fc_out's
// three HP-wide rows of a use_viewdirs = 0 network, or fc_rgb's three HPD-wide ones), then the two encoding argument tables.
__host__ __device__ constexpr int g_bias_floats(int nt, int layers) { return 16 * nt * (1 + layers) + 16 * ((nt + 1) / 2) + 4; }
__host__ __device__ constexpr int g_head_floats(int nt) { return 16 * nt + 3 * 16 * nt; }
struct GEncArg;
__host__ inline int g_lds_bytes(int ring_bytes, int nt, int layers, int enc_parts);      // defined below GEncArg
__host__ inline int g_bwd_lds_bytes(int ring_bytes, int nt) { return ring_bytes + g_head_floats(nt) * 4; }

// one encoding argument a < 3 F: coordinate a / F times frequency band a % F (modules.py:30-33, coordinate-major);
// a >= 3 F (the odd tail): band 0 -> sin 0 / cos 1 against zero weights
struct GEncArg { float band; int32_t coord; };
// (enc_parts: 1 for the one-part instantiations, G_ENC_PARTS for the LONG ones -- their tables hold both parts of both encodings)
__host__ inline int g_lds_bytes(int ring_bytes, int nt, int layers, int enc_parts) {
    return ring_bytes + (g_bias_floats(nt, layers) + g_head_floats(nt)) * 4 + 2 * enc_parts * G_ENC_ARGS * (int)sizeof(GEncArg);
}

// One GEMM stage of the generic kernel: acc[NT tiles] += W_stage * b, chunk by chunk through the 2-slot ring.
// On entry the stage's first chunk is resident in slot `par`; on exit the chunk (tail_src, tail_bytes) -- the first chunk of
// whatever runs next -- is resident in slot `par`.  RUNTIME: `nchunks` (>= 1) whole chunks of KCH k-steps are present
// (an encoding stage); otherwise the stage spans exactly KS k-steps (last chunk partial).
template <int NT, int KS, int NW, int KCH, bool RUNTIME>
__device__ __forceinline__ void gemm_stage_g(f32x4 (&acc)[NT], const float (&b)[KS], int nchunks, const char* gw,
                                             const char* tail_src, int tail_bytes, char* lds, int slot_bytes, int& par,
                                             int wave, int lane) {
    constexpr int NB = (NT + 3) / 4;
    constexpr int STEP_BYTES = NB * 1024;
    constexpr int NCH = (KS + KCH - 1) / KCH;
    static_assert(!RUNTIME || KS % KCH == 0, "an encoding stage is a whole number of chunks");
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        if constexpr (RUNTIME) {
            if (c >= nchunks) break;      // uniform
        }
        const int steps = (KS - c * KCH) < KCH ? (KS - c * KCH) : KCH;
        const int steps_next = (KS - (c + 1) * KCH) < KCH ? (KS - (c + 1) * KCH) : KCH;
        const bool last = RUNTIME ? (c + 1 >= nchunks) : (c + 1 == NCH);
        const char* next_src = last ? tail_src : gw + (c + 1) * KCH * STEP_BYTES;
        const int next_bytes = last ? tail_bytes : steps_next * STEP_BYTES;
        stream_to_lds<NW>(next_src, lds + (par ^ 1) * slot_bytes, next_bytes, wave, lane);
        const char* buf = lds + par * slot_bytes + lane * 16;
        const int nblk = steps * NB;      // blocks of this chunk: block j = (k-step j / NB, tiles 4 (j % NB) ..), 1 KiB apart
        f32x4 ab[3];
        ab[0] = *reinterpret_cast<const f32x4*>(buf);
        if (nblk > 1) ab[1] = *reinterpret_cast<const f32x4*>(buf + 1024);
#pragma unroll
        for (int j = 0; j < nblk; ++j) {
            if (j + 2 < nblk) ab[(j + 2) % 3] = *reinterpret_cast<const f32x4*>(buf + (j + 2) * 1024);
            __builtin_amdgcn_sched_barrier(0);
            const int ks = j / NB, blk = j % NB;
            const float bv = b[c * KCH + ks];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (blk * 4 + q < NT)
                    acc[blk * 4 + q] = __builtin_amdgcn_mfma_f32_16x16x4f32(ab[j % 3][q], bv, acc[blk * 4 + q], 0, 0, 0);
        }
        __syncthreads();   // this wave's DMA pieces have landed (vmcnt(0)); slot `par` is free for the next fill
        par ^= 1;
    }
}

// Positional encoding as MFMA B operands with the number of frequencies at run time (layout of mlp_device.h's encode: k-step
// s < ns carries arguments 2 s, 2 s + 1 -- lane groups sin(a0) cos(a0) sin(a1) cos(a1) --, then the identity step (x, y, z, 0)
// when the input itself is included, then zeros).  Same sincosf-halving exchange between a sin group and its cos group.
__device__ __forceinline__ void encode_g(float (&enc)[G_ENC_STEPS], const float (&x)[3], const GEncArg* tab, int ns,
                                         int ident, int g) {
    const int hi = g >> 1;
    const bool want_cos = (g & 1) != 0;
    auto argument = [&](int a) -> float {         // a: per-lane argument index
        const GEncArg e = tab[a];
        const float xv = e.coord == 0 ? x[0] : (e.coord == 1 ? x[1] : x[2]);
        return xv * e.band;
    };
    auto rest = [&](int t) -> float {             // the identity step, or padding
        return (ident && t == ns) ? (g == 0 ? x[0] : (g == 1 ? x[1] : (g == 2 ? x[2] : 0.0f))) : 0.0f;
    };
#pragma unroll
    for (int s = 0; s < G_ENC_STEPS; s += 2) {
        if (s + 1 < ns) {          // uniform: both k-steps carry arguments
            const float mine = argument(2 * (s + (want_cos ? 1 : 0)) + hi);
            float sv, cv;
            sincosf(mine, &sv, &cv);
            const float give = want_cos ? sv : cv;
            const float got = __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(give), 0x401f));   // lane ^ 16
            enc[s] = want_cos ? got : sv;
            enc[s + 1] = want_cos ? cv : got;
        } else if (s < ns) {       // the odd last argument step
            float sv, cv;
            sincosf(argument(2 * s + hi), &sv, &cv);
            enc[s] = want_cos ? cv : sv;
            enc[s + 1] = rest(s + 1);
        } else {
            enc[s] = rest(s);
            enc[s + 1] = rest(s + 1);
        }
    }
}

// The GEMM stages of one encoding's columns (layer1, a skip layer, layers_dir[0]): `ch` whole chunks of the stream, of which the
// first G_ENC_STEPS / KCH carry part 0 of the encoding (k-steps 0 .. 23: `enc0` if the caller holds it -- HAVE0 --, evaluated here
// otherwise) and the rest part 1 (k-steps 24 .. 47, more than 15 functions: evaluated here, into the same registers).  `ns` argument
// k-steps in all, `ident`: an identity step follows them; tab: this encoding's [G_ENC_PARTS][G_ENC_ARGS] table.  The two-part code is its
// own instantiation (LONG: the *_long translation units, plans of variant G_LONG_VARIANT): twelve more inlined sincosf sites per stage
// cost the one-part kernels ~10 registers and a few spills for nothing.
__device__ __forceinline__ int g_part_ns(int ns, int part) { const int r = ns - part * G_ENC_STEPS; return r < 0 ? 0 : (r > G_ENC_STEPS ? G_ENC_STEPS : r); }
__device__ __forceinline__ int g_part_ident(int ns, int ident, int part) { return ident && ns >= part * G_ENC_STEPS && ns < (part + 1) * G_ENC_STEPS; }

template <int NTO, int NW, int KCH, bool HAVE0, bool LONG>
__device__ __forceinline__ void enc_stages_g(f32x4 (&acc)[NTO], const float (&enc0)[HAVE0 ? G_ENC_STEPS : 1], const float (&x)[3],
                                             const GEncArg* tab, int ns, int ident, int ch, int g, const char* gw, const char* tail_src,
                                             int tail_bytes, char* lds, int slot_bytes, int& par, int wave, int lane) {
    if constexpr (!LONG) {          // one part: exactly the stage the kernels had before there were two
        if constexpr (HAVE0) {
            gemm_stage_g<NTO, G_ENC_STEPS, NW, KCH, true>(acc, enc0, ch, gw, tail_src, tail_bytes, lds, slot_bytes, par, wave, lane);
        } else {
            float e[G_ENC_STEPS];
            encode_g(e, x, tab, ns, ident, g);
            gemm_stage_g<NTO, G_ENC_STEPS, NW, KCH, true>(acc, e, ch, gw, tail_src, tail_bytes, lds, slot_bytes, par, wave, lane);
        }
    } else {
        constexpr int CPP = G_ENC_STEPS / KCH, STEP_BYTES = ((NTO + 3) / 4) * 1024;
        const int ch0 = ch < CPP ? ch : CPP, ch1 = ch - ch0;                    // uniform
        const char* mid = gw + ch0 * KCH * STEP_BYTES;
        if constexpr (HAVE0) {
            gemm_stage_g<NTO, G_ENC_STEPS, NW, KCH, true>(acc, enc0, ch0, gw, ch1 ? mid : tail_src, ch1 ? KCH * STEP_BYTES : tail_bytes, lds,
                                                          slot_bytes, par, wave, lane);
        } else {
            float e[G_ENC_STEPS];
            encode_g(e, x, tab, g_part_ns(ns, 0), g_part_ident(ns, ident, 0), g);
            gemm_stage_g<NTO, G_ENC_STEPS, NW, KCH, true>(acc, e, ch0, gw, ch1 ? mid : tail_src, ch1 ? KCH * STEP_BYTES : tail_bytes, lds,
                                                          slot_bytes, par, wave, lane);
        }
        if (ch1) {
            float e[G_ENC_STEPS];
            encode_g(e, x, tab + G_ENC_ARGS, g_part_ns(ns, 1), g_part_ident(ns, ident, 1), g);
            gemm_stage_g<NTO, G_ENC_STEPS, NW, KCH, true>(acc, e, ch1, mid, tail_src, tail_bytes, lds, slot_bytes, par, wave, lane);
        }
    }
}

// ---- training tape of the generic family: row-major [sample][width] rows of the REAL width (padding never leaves the
// registers).  16-byte accesses when the row stride allows it (width % 4 == 0), element-wise otherwise (e.g. the 50-wide view
// layer of a 100-wide network).
template <int NT>
__device__ __forceinline__ void store_rows_g(float* base, int width, int64_t sample, bool valid, const float (&op)[4 * NT], int g) {
    if (!valid) return;
    float* row = base + sample * width;
    const bool vec = (width & 3) == 0;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int k0 = 16 * nt + 4 * g;
        if (vec && k0 + 3 < width) {
            const f32x4 v = {op[4 * nt], op[4 * nt + 1], op[4 * nt + 2], op[4 * nt + 3]};
            *reinterpret_cast<f32x4*>(row + k0) = v;
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (k0 + r < width) row[k0 + r] = op[4 * nt + r];
        }
    }
}

// the same rows back into D layout (zeros beyond the real width): the backward pass takes ReLU' from the taped activations
template <int NT>
__device__ __forceinline__ void load_rows_g(const float* base, int width, int64_t sample, float (&op)[4 * NT], int g) {
    const float* row = base + sample * width;
    const bool vec = (width & 3) == 0;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int k0 = 16 * nt + 4 * g;
        if (vec && k0 + 3 < width) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(row + k0);
            op[4 * nt] = v[0]; op[4 * nt + 1] = v[1]; op[4 * nt + 2] = v[2]; op[4 * nt + 3] = v[3];
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) op[4 * nt + r] = k0 + r < width ? row[k0 + r] : 0.0f;
        }
    }
}

// TAPE (nm_mlp_forward_train on a generic-shape handle): the post-activations leave the registers once, as those rows
// (tape_h[0] = layer1's output, tape_h[1 + i] = relu(layers_xyz[i]), tape_feat, tape_v); no ReLU masks are written --
// the generic backward kernel reads the signs off the tape.  The radiance is the inference kernel's bit for bit.
// Occupancy: classes up to 10 tiles (hidden_size <= 160) are compiled for four waves per SIMD (128 registers; classes 6 -- 10
// spill 3 -- 88 registers for it, outside the k-step loops): +5 -- +6.5 points at 96 and 144 wide, +4.6 on the 8x128 shape, +1.2
// at 160; wider classes lose more to spills than the two extra waves hide.  The training kernels of these classes measured
// within +-3 % either way and simply share the bound.
template <int NT, int NW, int KCH, bool TAPE = false, bool LONG = false>
__global__ __launch_bounds__(NW * 64, NW == 8 ? (NT <= 10 ? 4 : 2) : 1) void mlp_kernel_g(const MlpArgs args, const int num_layers,
                                                                       const int density_only) {
    constexpr int HP = 16 * NT, NTD = (NT + 1) / 2, HPD = 16 * NTD;
    constexpr int KH = 4 * NT, KD = 4 * NTD;
    constexpr int NB = (NT + 3) / 4, NBD = (NTD + 3) / 4;
    constexpr int STEP = NB * 1024, STEPD = NBD * 1024;
    constexpr int SLOT = KCH * STEP;
    constexpr int FIRST_H = (KH < KCH ? KH : KCH) * STEP;       // first chunk of a hidden-input stage (trunk tiles)
    constexpr int FIRST_HD = (KH < KCH ? KH : KCH) * STEPD;     // ... of layers_dir[0]'s hidden columns
    extern __shared__ __attribute__((aligned(16))) char lds[];
    float* lds_bias = reinterpret_cast<float*>(lds + 2 * SLOT);
    const int nbias = g_bias_floats(NT, num_layers);            // layer1 | layers_xyz.* | fc_feat | layers_dir.0 | fc_alpha.b | fc_rgb.b[3]
    float* lds_walpha = lds_bias + nbias;                       // [4][HP / 4]
    float* lds_wrgb = lds_walpha + HP;                          // [3][4][HPD / 4], or [3][4][HP / 4] for a use_viewdirs = 0 network
    const bool flat = density_only == 2;
    const int nrgb = flat ? 3 * HP : 3 * HPD;
    constexpr int PARTS = LONG ? G_ENC_PARTS : 1;
    GEncArg* lds_tab = reinterpret_cast<GEncArg*>(lds_walpha + g_head_floats(NT));   // [xyz, dir][PARTS][G_ENC_ARGS]
    for (int i = threadIdx.x; i < nbias; i += NW * 64) lds_bias[i] = args.bias[i];
    for (int i = threadIdx.x; i < HP; i += NW * 64) lds_walpha[i] = args.walpha[i];
    for (int i = threadIdx.x; i < nrgb; i += NW * 64) lds_wrgb[i] = args.wrgb[i];
    for (int i = threadIdx.x; i < 2 * PARTS * G_ENC_ARGS; i += NW * 64) lds_tab[i] = static_cast<const GEncArg*>(args.g_tab)[i];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, col = lane & 15;
    const float* tail_bias = lds_bias + nbias - 4;
    const int chx = args.g_chx, chd = args.g_chd;               // whole chunks of the xyz / direction encoding stages
    const int enc_x_bytes = chx * KCH * STEP;                   // an xyz encoding stage in the stream (trunk tiles)

    const int64_t wg_iters = (args.n + NW * 16 - 1) / (NW * 16);
    int par = 0;
    if ((int64_t)blockIdx.x < wg_iters) stream_to_lds<NW>(args.wstream, lds, KCH * STEP, wave, lane);
    __syncthreads();   // tables, biases and layer1's first chunk are resident (later tiles: the previous tile's last stage fetched it)

    for (int64_t it = blockIdx.x; it < wg_iters; it += gridDim.x) {
        const bool has_next = it + gridDim.x < wg_iters;
        const int wrap_bytes = has_next ? KCH * STEP : 0;        // the next tile's first chunk: layer1's encoding columns
        const int64_t sample = (it * NW + wave) * 16 + col;
        const bool valid = sample < args.n;
        const int64_t sidx = valid ? sample : args.n - 1;
        const SamplePD smp = fetch_sample(args, sidx);
        const float p[3] = {smp.px, smp.py, smp.pz}, d[3] = {smp.dx, smp.dy, smp.dz};
        // classes 21 -- 24 at two waves per SIMD have no 24 registers to hold the encoding across the trunk: the skip layer
        // computes it again (the same values; `opaque` keeps the compiler from hoisting the second evaluation back up here)
        constexpr bool KEEP_ENC = !(NW == 8 && NT > 20);
        auto opaque = [](int v) { asm volatile("" : "+v"(v)); return v; };
        float encx[KEEP_ENC ? G_ENC_STEPS : 1];

        f32x4 acc[NT];
        float in[KH];
        const char* gw = args.wstream;
        // ---- layer1: xyz_enc -> H, no activation (models.py:62)
        load_bias<NT>(acc, lds_bias, g);
        if constexpr (KEEP_ENC) encode_g(encx, p, lds_tab, LONG ? g_part_ns(args.g_nsx, 0) : args.g_nsx, LONG ? g_part_ident(args.g_nsx, args.g_idx, 0) : args.g_idx, g);
        enc_stages_g<NT, NW, KCH, KEEP_ENC, LONG>(acc, encx, p, lds_tab, args.g_nsx, args.g_idx, chx, KEEP_ENC ? g : opaque(g), gw, gw + enc_x_bytes,
                                            FIRST_H, lds, SLOT, par, wave, lane);
        gw += enc_x_bytes;
        acc_to_operand<NT, false>(acc, in);
        if constexpr (TAPE) store_rows_g<NT>(args.tape_h, args.g_h, sample, valid, in, g);

        // ---- layers_xyz[0 .. L-2], then (full evaluation only) fc_feat as iteration L-1 (models.py:63-70)
        float sigma = 0.0f;
        const int trunk_iters = density_only ? num_layers - 1 : num_layers;
#pragma unroll 1
        for (int i = 0; i < trunk_iters; ++i) {
            const bool is_feat = i == num_layers - 1;
            if (is_feat) sigma = alpha_gemv<HP>(in, lds_walpha, g) + tail_bias[0];   // on the pre-feature activation
            const bool skip = !is_feat && ((args.skip_mask >> i) & 1u);
            const bool last_density = density_only && i == num_layers - 2;
            load_bias<NT>(acc, lds_bias + HP * (1 + i), g);
            {
                const char* after = gw + KH * STEP;
                const char* tsrc = after;
                int tbytes = FIRST_H;
                if (skip) tbytes = KCH * STEP;                    // the skip layer's encoding columns follow
                else if (is_feat) tbytes = FIRST_HD;              // layers_dir[0]'s hidden columns follow
                else if (last_density) { tsrc = args.wstream; tbytes = wrap_bytes; }
                gemm_stage_g<NT, KH, NW, KCH, false>(acc, in, 0, gw, tsrc, tbytes, lds, SLOT, par, wave, lane);
                gw = after;
            }
            if (skip) {   // cat(hidden, xyz_enc): the encoding columns of layers_xyz[i] (models.py:64-65)
                const char* after = gw + enc_x_bytes;
                const char* tsrc = after;
                int tbytes = FIRST_H;
                if (last_density) { tsrc = args.wstream; tbytes = wrap_bytes; }
                enc_stages_g<NT, NW, KCH, KEEP_ENC, LONG>(acc, encx, p, lds_tab, args.g_nsx, args.g_idx, chx, opaque(g), gw, tsrc, tbytes, lds, SLOT,
                                                    par, wave, lane);
                gw = after;
            }
            acc_to_operand<NT, true>(acc, in);
            if constexpr (TAPE) {
                float* dst = is_feat ? args.tape_feat : args.tape_h + (int64_t)(1 + i) * args.n * args.g_h;
                store_rows_g<NT>(dst, args.g_h, sample, valid, in, g);
            }
        }

        if (density_only) {
            sigma = alpha_gemv<HP>(in, lds_walpha, g) + tail_bias[0];
            if (flat) flat_head<HP>(args, in, lds_wrgb, tail_bias, sigma, sample, valid, g);   // use_viewdirs = 0 (models.py:77-79)
            else if (valid && g == 0) args.out[sample] = sigma;
            continue;
        }

        // ---- layers_dir[0]: cat(feat, dir_enc) -> H/2, relu (models.py:72-74): hidden columns, then the encoding columns
        f32x4 accd[NTD];
        float v[KD];
        load_bias<NTD>(accd, lds_bias + HP * (1 + num_layers), g);
        {
            const char* after = gw + KH * STEPD;
            const bool has_enc = chd > 0;
            gemm_stage_g<NTD, KH, NW, KCH, false>(accd, in, 0, gw, has_enc ? after : args.wstream,
                                                  has_enc ? KCH * STEPD : wrap_bytes, lds, SLOT, par, wave, lane);
            gw = after;
            if (has_enc) {
                const float none[1] = {0.0f};
                enc_stages_g<NTD, NW, KCH, false, LONG>(accd, none, d, lds_tab + PARTS * G_ENC_ARGS, args.g_nsd, args.g_idd, chd, g, gw,
                                                  args.wstream, wrap_bytes, lds, SLOT, par, wave, lane);
            }
        }
        acc_to_operand<NTD, true>(accd, v);
        if constexpr (TAPE) store_rows_g<NTD>(args.tape_v, args.g_hd, sample, valid, v, g);

        // ---- fc_rgb + sigmoid (models.py:75), 3-row GEMV on the VALU
        float rgb[3];
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
        if (valid && g == 0) {
            f32x4 o4 = {rgb[0], rgb[1], rgb[2], sigma};
            *reinterpret_cast<f32x4*>(args.out + 4 * sample) = o4;
        }
    }
}


// ---- delta propagation of the generic family (nm_mlp_backward on a generic-shape handle): what nerf_train.hip's
// mlp_backward_kernel does for the tuned shapes -- the transposed layers in reverse order on the same register-resident chain
// (hidden columns only; the encodings have no gradient) -- with the padded width classes of this header, ReLU' read off the
// taped activations (a zero activation passes no gradient: the subgradient autograd uses) and each delta stored as rows of
// the real width once its stage is done.  Stream: [layers_dir.0^T | fc_feat^T |] layers_xyz[L-2 .. 0]^T (mlp_api.hip).
template <int NT>
__device__ __forceinline__ void relu_gate(const f32x4 (&acc)[NT], const float* base, int width, int64_t sample, float (&op)[4 * NT], int g) {
    const float* row = base + sample * width;      // the taped activation of this lane's sample, read tile by tile (no second register array)
    const bool vec = (width & 3) == 0;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int k0 = 16 * nt + 4 * g;
        f32x4 a = {0.0f, 0.0f, 0.0f, 0.0f};
        if (vec && k0 + 3 < width) a = *reinterpret_cast<const f32x4*>(row + k0);
        else {
#pragma unroll
            for (int r = 0; r < 4; ++r) a[r] = k0 + r < width ? row[k0 + r] : 0.0f;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) op[4 * nt + r] = a[r] > 0.0f ? acc[nt][r] : 0.0f;
    }
}

template <int NT, int NW, int KCH>
__global__ __launch_bounds__(NW * 64, NW == 8 ? (NT <= 10 ? 4 : 2) : 1) void mlp_backward_kernel_g(const MlpBwdArgs args, const int num_layers,
                                                                                const int flat) {
    constexpr int HP = 16 * NT, NTD = (NT + 1) / 2, HPD = 16 * NTD;
    constexpr int KH = 4 * NT, KD = 4 * NTD;
    constexpr int NB = (NT + 3) / 4;
    constexpr int STEP = NB * 1024, SLOT = KCH * STEP;
    constexpr int FIRST_D = (KD < KCH ? KD : KCH) * STEP;       // first chunk of layers_dir.0^T (KD k-steps onto the trunk tiles)
    constexpr int FIRST_H = (KH < KCH ? KH : KCH) * STEP;       // ... of a hidden^T stage
    extern __shared__ __attribute__((aligned(16))) char lds[];
    float* lds_walpha = reinterpret_cast<float*>(lds + 2 * SLOT);   // [4][HP / 4]
    float* lds_wrgb = lds_walpha + HP;                              // [3][4][HPD / 4], or (flat) [3][4][HP / 4]
    for (int i = threadIdx.x; i < HP; i += NW * 64) lds_walpha[i] = args.walpha[i];
    for (int i = threadIdx.x; i < (flat ? 3 * HP : 3 * HPD); i += NW * 64) lds_wrgb[i] = args.wrgb[i];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, col = lane & 15;
    const int L = num_layers, H = args.g_h, HD = args.g_hd;
    const int first_bytes = flat ? FIRST_H : FIRST_D;

    const int64_t wg_iters = (args.n + NW * 16 - 1) / (NW * 16);
    int par = 0;
    if ((int64_t)blockIdx.x < wg_iters) stream_to_lds<NW>(args.wstream, lds, first_bytes, wave, lane);
    __syncthreads();

    for (int64_t it = blockIdx.x; it < wg_iters; it += gridDim.x) {
        const bool has_next = it + gridDim.x < wg_iters;
        const int wrap_bytes = has_next ? first_bytes : 0;
        const int64_t sample = (it * NW + wave) * 16 + col;
        const bool valid = sample < args.n;
        const int64_t sidx = valid ? sample : args.n - 1;
        // ---- head: sigmoid' (models.py:75 / :78)
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
        f32x4 acc[NT];
        float in[KH];
        const char* gw = args.wstream;
        if (flat) {
            // ---- fc_out^T on the VALU: delta at the trunk's output from the four head deltas (rows in fc_alpha's operand layout)
            const float* wa = lds_walpha + g * (HP / 4);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                f32x4 w4 = *reinterpret_cast<const f32x4*>(wa + 4 * nt);
                f32x4 a = {w4[0] * dsigma, w4[1] * dsigma, w4[2] * dsigma, w4[3] * dsigma};
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) {
                    w4 = *reinterpret_cast<const f32x4*>(lds_wrgb + ch * HP + g * (HP / 4) + 4 * nt);
#pragma unroll
                    for (int r = 0; r < 4; ++r) a[r] = fmaf(w4[r], drgb[ch], a[r]);
                }
                acc[nt] = a;
            }
        } else {
            // ---- fc_rgb^T on the VALU, gated by the view layer's ReLU: delta at layers_dir.0's pre-activation
            float dv[KD], av[KD];
            load_rows_g<NTD>(args.tape_v, HD, sidx, av, g);
#pragma unroll
            for (int s = 0; s < KD; ++s) {
                float a = lds_wrgb[(0 * 4 + g) * KD + s] * drgb[0];
                a = fmaf(lds_wrgb[(1 * 4 + g) * KD + s], drgb[1], a);
                a = fmaf(lds_wrgb[(2 * 4 + g) * KD + s], drgb[2], a);
                dv[s] = av[s] > 0.0f ? a : 0.0f;
            }
            store_rows_g<NTD>(args.d_v, HD, sample, valid, dv, g);
            // ---- layers_dir.0^T (hidden columns): -> delta at relu(fc_feat) -> gated -> delta at fc_feat's output
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            gemm_stage_g<NT, KD, NW, KCH, false>(acc, dv, 0, gw, gw + KD * STEP, FIRST_H, lds, SLOT, par, wave, lane);
            gw += KD * STEP;
            relu_gate<NT>(acc, args.tape_feat, H, sidx, in, g);
            store_rows_g<NT>(args.d_feat, H, sample, valid, in, g);
            // ---- fc_feat^T + fc_alpha^T: delta at the output of layers_xyz[L-2]
            const float* wa = lds_walpha + g * (HP / 4);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const f32x4 w4 = *reinterpret_cast<const f32x4*>(wa + 4 * nt);
                acc[nt] = f32x4{w4[0] * dsigma, w4[1] * dsigma, w4[2] * dsigma, w4[3] * dsigma};
            }
            gemm_stage_g<NT, KH, NW, KCH, false>(acc, in, 0, gw, gw + KH * STEP, FIRST_H, lds, SLOT, par, wave, lane);
            gw += KH * STEP;
        }
        relu_gate<NT>(acc, args.tape_h + (int64_t)(L - 1) * args.n * H, H, sidx, in, g);
        store_rows_g<NT>(args.d_h + (int64_t)(L - 1) * args.n * H, H, sample, valid, in, g);
        // ---- layers_xyz[i]^T, i = L-2 .. 0: delta at the input of layers_xyz[i] (gated by the ReLU of layers_xyz[i-1]; layer1 has none)
#pragma unroll 1
        for (int i = L - 2; i >= 0; --i) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            const char* after = gw + KH * STEP;
            gemm_stage_g<NT, KH, NW, KCH, false>(acc, in, 0, gw, i == 0 ? args.wstream : after, i == 0 ? wrap_bytes : FIRST_H, lds,
                                                 SLOT, par, wave, lane);
            gw = after;
            if (i > 0) {
                relu_gate<NT>(acc, args.tape_h + (int64_t)i * args.n * H, H, sidx, in, g);
            } else {
                acc_to_operand<NT, false>(acc, in);
            }
            store_rows_g<NT>(args.d_h + (int64_t)i * args.n * H, H, sample, valid, in, g);
        }
    }
}

}  // namespace nm
