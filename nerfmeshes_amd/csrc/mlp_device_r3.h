// Round-2 inference kernel: the same dataflow as mlp_device.h's mlp_kernel (register-resident activations, weights
// as the only streamed operand, bit-identical results) with the two stalls the round-1 ablations located removed:
//
//   * 3-slot LDS ring, weight DMA TWO chunks ahead.  In the 2-slot kernel the chunk that is read next only becomes
//     visible at the barrier that ends the current chunk, so the first A-operand ds_read of every chunk is issued AFTER
//     the barrier and BOTH waves of a SIMD -- released together -- expose its latency together (plus the DMA issue
//     sequence in front of it): ~300 idle matrix-pipe cycles per 8192-cycle chunk.  With the chunk after next in
//     flight, the next chunk is already resident while the current one is consumed, and the operand prefetch runs
//     ACROSS chunk and stage boundaries (`carry`): after a barrier the first MFMA issues immediately.
//   * the DMA issue of the two waves sharing a SIMD is staggered (waves 0..NW/2-1 at k-step 0, the others half a chunk
//     later): one wave's scalar/VMEM issue sequence runs under its partner's MFMAs instead of next to the partner's
//     identical sequence.
// Measured (profiles/r02_mlp_variants.json): on their own these two are worth +0.1 % -- the stall that mattered was
// the address-VGPR read of the DMA instruction itself (see stream_to_lds in mlp_device.h).
#pragma once
#include "mlp_device.h"

namespace nm {

struct NextChunks {          // the first two chunks of whatever stage runs next (bytes 0: nothing follows)
    const char* s0; int b0;
    const char* s1; int b1;
};

// A chunk is consumed as a stream of "blocks": block j = one ds_read_b128 per lane (the A operands of 4 consecutive
// output tiles of k-step j / NB) feeding 4 MFMAs; blocks are contiguous in the chunk image (offset j * 1 KiB).  The
// operands of block j + 2 are fetched while block j runs (8 MFMAs = 256 matrix-pipe cycles of cover, 3 live operand
// quads instead of round 1's 8), and the stream simply continues into the next chunk / the next stage: `carry` holds
// blocks 0 and 1 of whatever comes next.
// STAG: 0 every wave issues its DMA pieces at block 0 of a chunk | 1 waves 0..NW/2-1 at block 0, the others half a chunk
// later.  ABL (timing-only ablations, ablation library): 1 barriers do not wait for the DMA | 2 no DMA | 4 one lane per
// DMA instruction | 8 round-1 DMA form (global_load_lds with per-lane address VGPRs).
// STORE (training): b1 is written to `store_row` a few tiles per chunk while the stage consumes it (see gemm_stage).
template <int NT, int KS1, int KS2, int NW, int LDSBUF, int KCH, int STAG, int ABL = 0, bool STORE = false>
__device__ __forceinline__ void gemm_stage3(f32x4 (&acc)[NT], const float (&b1)[KS1],
                                            const float (&b2)[(KS2 > 0 ? KS2 : 1)], const char* gw,
                                            const NextChunks nx, char* lds, int& slot, f32x4 (&carry)[2],
                                            int wave, int lane, float* store_row = nullptr) {
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
#pragma unroll
        for (int j = 0; j < nblk; ++j) {
            auto dma = [&]() {
                if constexpr (ABL & 8) stream_to_lds_vaddr<NW>(src, dst, bytes, wave, lane);
                else stream_to_lds<NW>(src, dst, bytes, wave, lane);
            };
            if constexpr (ABL & 2) {
            } else if constexpr (ABL & 4) {   // timing only: the same instruction sequence moving 16 B instead of 1 KiB per piece
                if (lane == 0) {
                    if (j == 0 && wave < NW / 2) stream_to_lds_vaddr<NW>(src, dst, bytes, wave, lane);
                    if (j == nblk / 2 && wave >= NW / 2) stream_to_lds_vaddr<NW>(src, dst, bytes, wave, lane);
                }
            } else if constexpr (STAG == 1) {
                if (j == 0 && wave < NW / 2) dma();
                if (j == nblk / 2 && wave >= NW / 2) dma();
            } else {
                if (j == 0) dma();
            }
            const int ks = j / NB, blk = j % NB;
            const int s = c * KCH + ks;
            const float b = s < KS1 ? b1[s < KS1 ? s : 0] : b2[s >= KS1 ? s - KS1 : 0];
            // the chunk after this one is resident since the last barrier: the stream runs straight into it
            const char* from = (j + 2 < nblk) ? buf + (j + 2) * 1024 : nbuf + (j + 2 - nblk) * 1024;
            const int r0 = (c * KCH * NB + j) % 3;           // ring position of block j (static: loops are unrolled)
            ab[(r0 + 2) % 3] = *reinterpret_cast<const f32x4*>(from);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                acc[blk * 4 + q] = __builtin_amdgcn_mfma_f32_16x16x4f32(ab[r0][q], b, acc[blk * 4 + q], 0, 0, 0);
        }
        if constexpr (ABL & 1) __builtin_amdgcn_s_barrier();   // timing-only: no wait for the DMA (results WRONG)
        else __syncthreads();   // this wave's DMA pieces have landed (vmcnt(0)); after it the chunk after next is visible to all
        slot = slot1;
    }
    constexpr int TOTAL = KS * NB;                           // blocks consumed: the two in flight sit at TOTAL, TOTAL + 1
    carry[0] = ab[TOTAL % 3];
    carry[1] = ab[(TOTAL + 1) % 3];
}

// TAPE (EXPERIMENT, instantiated in the ablation library only -- nerf_train.hip, NM_MLP_VARIANT=3): the taping forward on this
// dataflow, writing the same tape as mlp_kernel<..., TAPE> (activation rows while the next stage consumes them, ReLU bit masks
// per tile), the same bits.  Measured 4.7 % SLOWER than the 2-slot taping kernel (3.54 vs 3.38 ms per 393 216 samples of the
// 8x256 network, profiles/r04_train_three_slot.json): the training kernels stay on mlp_device.h's dataflow.
template <int H, int FX, int FD, int NW, int KCH, int STAG, int ABL = 0, bool FLAT = false, bool TAPE = false>   // FLAT: see mlp_kernel
// Occupancy: networks up to 128 wide are compiled for FOUR waves per SIMD (128 registers: two 8-wave workgroups per CU; the
// 128-wide instances spill 9 -- 17 registers outside the k-step loops for it).  With VALU issue time adding to matrix time on
// narrow networks two more waves per SIMD are worth +2.7 points at 8x128 (0.875 -> 0.902, same bits).
__global__ __launch_bounds__(NW * 64, (H <= 128 && !TAPE) ? 4 : 2) void mlp_kernel3(const MlpArgs args, const int num_layers,
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
    for (int i = threadIdx.x; i < ((FLAT && density_only == 2) ? 3 * H : 3 * H / 2); i += NW * 64) lds_wrgb[i] = args.wrgb[i];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, col = lane & 15;
    const float* tail_bias = lds_bias + nbias - 4;

    const int64_t wg_iters = (args.n + NW * 16 - 1) / (NW * 16);
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
        const int64_t sample = (it * NW + wave) * 16 + col;
        const bool valid = sample < args.n;
        const int64_t sidx = valid ? sample : args.n - 1;
        const SamplePD smp = fetch_sample(args, sidx);
        const float p[3] = {smp.px, smp.py, smp.pz}, d[3] = {smp.dx, smp.dy, smp.dz};
        const float dummy[1] = {0.0f};
        float encx[N::EX];
        encode<FX, N::EX, 0>(encx, p, args.bands_xyz, g);
        const NextChunks wrap = enc_next(args.wstream, has_next);

        f32x4 acc[N::NT];
        float in[N::KH];
        const char* gw = args.wstream;
        // ---- layer1: xyz_enc -> H, no activation (models.py:62)
        load_bias<N::NT>(acc, lds_bias, g);
        gemm_stage3<N::NT, N::EX, 0, NW, N::LDSBUF, KCH, STAG, ABL>(acc, encx, dummy, gw, hidden_next(gw + N::EX * N::STEP), lds,
                                                              slot, carry, wave, lane);
        gw += N::EX * N::STEP;
        acc_to_operand<N::NT, false>(acc, in);
        const int64_t tile = it * NW + wave;
        float* tape_row = (TAPE && valid) ? args.tape_h + sample * H + 4 * g : nullptr;

        // ---- layers_xyz[0 .. L-2], then (full evaluation only) fc_feat as iteration L-1 (models.py:63-70)
        float sigma = 0.0f;
        const int trunk_iters = density_only ? num_layers - 1 : num_layers;
#pragma unroll 1
        for (int i = 0; i < trunk_iters; ++i) {
            const bool is_feat = i == num_layers - 1;
            if (is_feat) sigma = alpha_gemv<H>(in, lds_walpha, g) + tail_bias[0];
            const bool skip = !is_feat && ((args.skip_mask >> i) & 1u);
            const bool last_density = density_only && i == num_layers - 2;
            load_bias<N::NT>(acc, lds_bias + H * (1 + i), g);
            {
                const char* after = gw + N::KH * N::STEP;
                NextChunks nx = hidden_next(after);
                if (skip) nx = enc_next(after, true);
                else if (is_feat) nx = dir_next(after);
                else if (last_density) nx = wrap;
                gemm_stage3<N::NT, N::KH, 0, NW, N::LDSBUF, KCH, STAG, ABL, TAPE>(acc, in, dummy, gw, nx, lds, slot, carry, wave, lane,
                                                                                 tape_row ? tape_row + (int64_t)i * args.n * H : nullptr);
                gw = after;
            }
            if (skip) {  // cat(hidden, xyz_enc): the encoding columns of layers_xyz[i] (models.py:64-65)
                const char* after = gw + N::EX * N::STEP;
                const NextChunks nx = last_density ? wrap : hidden_next(after);
                gemm_stage3<N::NT, N::EX, 0, NW, N::LDSBUF, KCH, STAG, ABL>(acc, encx, dummy, gw, nx, lds, slot, carry, wave, lane);
                gw = after;
            }
            acc_to_operand<N::NT, true>(acc, in);
            if constexpr (TAPE) {
                if (tile < args.tiles) args.mask_h[((int64_t)i * args.tiles + tile) * 64 + lane] = positive_mask(in);
            }
        }

        if (density_only) {
            sigma = alpha_gemv<H>(in, lds_walpha, g) + tail_bias[0];
            if (FLAT && density_only == 2) {   // use_viewdirs = 0
                flat_head<H>(args, in, lds_wrgb, tail_bias, sigma, sample, valid, g);
                // TAPE: the trunk's last activation has no stage behind it that would write it while consuming it
                if constexpr (TAPE) store_rows<N::NT>(args.tape_h + (int64_t)(num_layers - 1) * args.n * H, H, sample, valid, in, g);
            } else if (valid && g == 0) args.out[sample] = sigma;
            continue;
        }

        // ---- layers_dir[0]: cat(feat, dir_enc) -> H/2, relu (models.py:72-74)
        f32x4 accd[N::NTD];
        float v[N::KD];
        load_bias<N::NTD>(accd, lds_bias + H * (1 + num_layers), g);
        float encd[N::ED];
        encode<FD, N::ED, 0>(encd, d, args.bands_dir, g);
        gemm_stage3<N::NTD, N::KH, N::ED, NW, N::LDSBUF, KCH, STAG, ABL, TAPE>(accd, in, encd, gw, wrap, lds, slot, carry, wave, lane,
                                                                             (TAPE && valid) ? args.tape_feat + sample * H + 4 * g : nullptr);
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
            const float* wr = lds_wrgb + (ch * 4 + g) * N::KD;
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
