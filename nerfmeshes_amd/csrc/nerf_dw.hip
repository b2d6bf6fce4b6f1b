// Weight gradients of the fused MLP for gfx950: dW[out][in] = sum_n delta[n][out] * act[n][in] and the bias gradients
// db[out] = sum_n delta[n][out], for one (delta, activation) pair of tape rows -- what autograd's addmm backward does
// for every Linear of /root/reference/src/nerf/models.py:60-80 (round 1 ran these through rocBLAS with a manual split-K
// plus separate column sums: 35-45 % of a training iteration).
//
// Shape of the problem: a GEMM with a tiny output (<= 256 x 320) and a huge contraction (n = rays * samples ~ 4e5).
// One workgroup per CU takes a contiguous range of samples and accumulates a FULL-SIZE partial dW in registers
// (fp32 MFMA v_mfma_f32_16x16x4_f32: the contraction index of the instruction is the sample index); a second,
// order-fixed pass sums the per-workgroup partials -- deterministic, no floating-point atomics.
//
// Dataflow (mirrors the forward kernel): the row-major tape rows are the streamed operands.  16 samples of delta and of
// activation rows (16 KiB + 16 KiB for 256-wide rows) form a chunk, DMA'd HBM -> LDS THREE chunks ahead into a 4-slot
// ring by scalar-addressed buffer_load ... lds (mlp_device.h: stream_to_lds); one barrier per chunk, in front of it a
// COUNTED s_waitcnt vmcnt(pieces of the newest chunk): unlike the forward kernel's weights these operands come from
// HBM, and a chunk (3.4 us of MFMAs) is not enough to cover that latency under load -- with vmcnt(0) at every barrier
// (3-slot ring, first version) the kernel ran at 120 TFLOP/s.  From LDS every
// lane fetches with ONE ds_read_b128 the operands of 4 output tiles: lane (k = l >> 4, i = l & 15) reads 4 consecutive
// features 4i .. 4i+3 of sample 4*ks + k of a 64-feature block, i.e. register q is the MFMA operand of "tile q" whose
// row/column i stands for feature 4i + q (the permutation is undone when the partial is written).  A wave owns a
// 64 x 128 block of dW (2 blocks of 64 x 64 = 32 tiles, 128 accumulator registers): 3 ds_read_b128 per 32 MFMAs.
// Narrow products (a 64-wide encoding operand) are split over the sample index between the wave halves instead.
// The bias gradient is a VALU by-product: the wave that owns the first column block adds its A operands up per lane.
//
// Roofline: MFMA (2 * out * in FLOP per sample); HBM traffic 4 * (out + in) B per sample = 64 FLOP/B for 256 x 256,
// i.e. 2.4 TB/s at the fp32 MFMA peak -- the two operand streams are read exactly once.
#include <cstdlib>

#include "nm_internal.h"
#include "mlp_device.h"

namespace nm {

struct DwArgs {
    const float* a;          // delta rows (n_pad, AB * 64), row-major
    const float* b;          // activation rows (n_pad, BB * 64), row-major
    int64_t chunks;          // n_pad / 16
    float* partial;          // (parts, AB * 64, BB * 64)
    float* partial_bias;     // (parts, AB * 64)
};

// Several products of ONE shape in one launch (round 5: the L same-shape layers of a network): workgroup x serves job
// x / per_job, sample part x % per_job.  With J jobs a job gets 1 / J of the CUs and J times the samples per workgroup: the
// same matrix work, but J times fewer partials to write and to reduce (8 x 256^2 layers: 64 MB instead of 512 MB per network)
// and one launch + one reduction instead of J of each.
constexpr int DW_MAX_JOBS = 16;
struct DwBatch {
    DwArgs job[DW_MAX_JOBS];
    int32_t jobs, per_job;
};

constexpr int DW_ROWS = 16;  // samples per chunk of the base geometry (4 k-groups of 4)

// ROWS: samples per chunk.  16 for the 256-wide products (32 KiB per chunk).  The narrower ones take 32: a chunk is what ONE
// barrier interval computes on, and the DMA of chunk c + 2 has two such intervals to land -- at 16 rows of a 128 x 128 product
// that is 2 x 0.85 us, less than the HBM latency under load (measured: tests/tools/probes/dma_ring.hip, 3-slot vs 4-slot leads);
// 32 rows double the lead at the same bytes in flight per barrier.
template <int AB, int BB, int ROWS = DW_ROWS>
__global__ __launch_bounds__(512, 2) void dw_kernel(const DwBatch batch) {
    const int job_i = blockIdx.x / batch.per_job;
    const int part_i = blockIdx.x - job_i * batch.per_job;
    const DwArgs args = batch.job[job_i];
    constexpr int NW = 8;
    constexpr int AW = AB * 64, BW = BB * 64;
    constexpr int A_BYTES = ROWS * AW * 4, B_BYTES = ROWS * BW * 4, SLOT = A_BYTES + B_BYTES;
    constexpr int NBLK = AB * BB;
    constexpr int BPW = NBLK >= NW ? NBLK / NW : 1;       // 64 x 64 blocks per wave
    constexpr int KSPLIT = NBLK >= NW ? 1 : NW / NBLK;    // wave groups sharing a block, split over the k-groups
    static_assert(NBLK >= NW ? NBLK % NW == 0 : NW % NBLK == 0, "block / wave mapping");
    static_assert(KSPLIT <= ROWS / 4 && BPW <= 2, "unsupported shape");
    static_assert(BPW == 1 || BB % BPW == 0, "a wave's blocks share their A block");
    static_assert(ROWS % 16 == 0, "whole k-groups for every k-split");
    constexpr int KG = (ROWS / 4) / KSPLIT;               // k-groups of a chunk this wave processes
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, i = lane & 15;
    const int id0 = NBLK >= NW ? wave * BPW : wave % NBLK;
    const int kpart = NBLK >= NW ? 0 : wave / NBLK;
    const int ablk = id0 / BB, bblk0 = id0 % BB;

    const int64_t c_lo = args.chunks * part_i / batch.per_job, c_hi = args.chunks * (part_i + 1) / batch.per_job;
    f32x4 acc[BPW][4][4];
#pragma unroll
    for (int p = 0; p < BPW; ++p)
#pragma unroll
        for (int qa = 0; qa < 4; ++qa)
#pragma unroll
            for (int qb = 0; qb < 4; ++qb) acc[p][qa][qb] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    f32x4 bias = {0.0f, 0.0f, 0.0f, 0.0f};

    // every wave issues the SAME number of 1 KiB pieces per chunk (the barrier's s_waitcnt counts them): operands narrower
    // than NW KiB per chunk are covered by letting the surplus waves repeat a piece (same bytes to the same place)
    auto even_pieces = [&](const char* src, char* dst, int bytes) {
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), (short)16, 0x7fffffff, 1 << 23);
        const int units = bytes >> 10;
        for (int u = wave; u < ((units + NW - 1) / NW) * NW; u += NW) {
            const int piece = u % units;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(dst + piece * 1024), 16, 0,
                                                     piece * 1024, 0, 0);
        }
    };
    auto dma = [&](int64_t c, int slot) {                 // chunk c -> ring slot (both operands)
        if (c >= c_hi) return;
        char* dst = lds + slot * SLOT;
        even_pieces(reinterpret_cast<const char*>(args.a) + c * A_BYTES, dst, A_BYTES);
        even_pieces(reinterpret_cast<const char*>(args.b) + c * B_BYTES, dst + A_BYTES, B_BYTES);
    };
    // operand addresses inside a slot: k-group ks, this lane's sample 4 ks + g
    const int a_off = g * (AW * 4) + ablk * 256 + i * 16;
    const int b_off = A_BYTES + g * (BW * 4) + bblk0 * 256 + i * 16;

    constexpr int PIECES = (A_BYTES / 1024 + NW - 1) / NW + (B_BYTES / 1024 + NW - 1) / NW;   // DMA instructions per wave per chunk
    static_assert(PIECES >= 2 && PIECES <= 8, "counted wait below");
    dma(c_lo, 0);
    dma(c_lo + 1, 1);
    dma(c_lo + 2, 2);
    __syncthreads();
    int slot = 0;
    f32x4 a_next, b_next[BPW];
    auto load_ops = [&](int s, int kg, f32x4& a, f32x4 (&b)[BPW]) {
        const int ks = kg * KSPLIT + kpart;                                   // k-group inside the chunk
        const char* pa = lds + s * SLOT + a_off + ks * 4 * (AW * 4);
        const char* pb = lds + s * SLOT + b_off + ks * 4 * (BW * 4);
        a = *reinterpret_cast<const f32x4*>(pa);
#pragma unroll
        for (int p = 0; p < BPW; ++p) b[p] = *reinterpret_cast<const f32x4*>(pb + p * 256);
    };
    if (c_lo < c_hi) load_ops(0, 0, a_next, b_next);

    for (int64_t c = c_lo; c < c_hi; ++c) {
        const int slot1 = (slot + 1) & 3;
        const int slot3 = (slot + 3) & 3;
#pragma unroll
        for (int kg = 0; kg < KG; ++kg) {
            // weight-stream style staggering: the two waves of a SIMD issue their DMA half a chunk apart
            if (kg == 0 && wave < NW / 2) dma(c + 3, slot3);
            if (kg == (KG > 1 ? KG / 2 : 0) && wave >= NW / 2) dma(c + 3, slot3);
            const f32x4 a = a_next;
            f32x4 b[BPW];
#pragma unroll
            for (int p = 0; p < BPW; ++p) b[p] = b_next[p];
            if (kg + 1 < KG) load_ops(slot, kg + 1, a_next, b_next);
            else load_ops(slot1, 0, a_next, b_next);         // next chunk: visible since the last barrier
            __builtin_amdgcn_sched_barrier(0);
            if (bblk0 == 0) bias += a;                       // column sums of delta (wave-uniform branch)
#pragma unroll
            for (int qa = 0; qa < 4; ++qa)
#pragma unroll
                for (int p = 0; p < BPW; ++p)
#pragma unroll
                    for (int qb = 0; qb < 4; ++qb)
                        acc[p][qa][qb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[qa], b[p][qb], acc[p][qa][qb], 0, 0, 0);
        }
        // chunk c + 2 (issued during chunk c - 1) must have landed before anybody reads it (from the end of chunk
        // c + 1 on); chunk c + 3, issued during this chunk, may stay in flight across the barrier
        if (c + 3 < c_hi) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        slot = slot1;
    }

    // ---- this workgroup's partial (feature permutation undone: tile (qa, qb) row i' = 4g + r, column j' = i is
    //      dW[64 ablk + 4 i' + qa][64 bblk + 4 j' + qb]).  Waves that split the k-groups of a block between them (KSPLIT > 1) first
    //      add their accumulators up through the LDS the ring no longer needs, in k-part order 0, 1, ... (deterministic): ONE partial
    //      per workgroup instead of KSPLIT -- the 64 x 64 kernel's 8 would otherwise make the order-fixed reduction read 8 x as much
    //      as the operands of a short product are worth (config 1: a 32 us reduction behind a 50 us product).
    float* out = args.partial + (int64_t)part_i * (AW * BW);
    float* out_bias = args.partial_bias + (int64_t)part_i * AW;
    float* red = reinterpret_cast<float*>(lds);                    // [KSPLIT][AW * BW] then [KSPLIT][AW]
    float* dst = KSPLIT > 1 ? red + kpart * (AW * BW) : out;
#pragma unroll
    for (int p = 0; p < BPW; ++p)
#pragma unroll
        for (int qa = 0; qa < 4; ++qa)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 64 * ablk + 4 * (4 * g + r) + qa;
                const f32x4 v = {acc[p][qa][0][r], acc[p][qa][1][r], acc[p][qa][2][r], acc[p][qa][3][r]};
                *reinterpret_cast<f32x4*>(dst + (int64_t)row * BW + 64 * (bblk0 + p) + 4 * i) = v;
            }
    if (bblk0 == 0) {
        // lanes (g, i) hold the sums over samples = g (mod 4) of features 4i .. 4i+3: fold the 4 lane groups
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float v = bias[q];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            bias[q] = v;
        }
        float* bdst = KSPLIT > 1 ? red + KSPLIT * (AW * BW) + kpart * AW : out_bias;
        if (g == 0) *reinterpret_cast<f32x4*>(bdst + 64 * ablk + 4 * i) = bias;
    }
    if constexpr (KSPLIT > 1) {
        __syncthreads();
        const f32x4* r4 = reinterpret_cast<const f32x4*>(red);
        for (int e = threadIdx.x; e < AW * BW / 4; e += NW * 64) {
            f32x4 sum = r4[e];
#pragma unroll
            for (int k = 1; k < KSPLIT; ++k) sum += r4[k * (AW * BW / 4) + e];
            reinterpret_cast<f32x4*>(out)[e] = sum;
        }
        const float* rb = red + KSPLIT * (AW * BW);
        for (int e = threadIdx.x; e < AW; e += NW * 64) {
            float sum = rb[e];
#pragma unroll
            for (int k = 1; k < KSPLIT; ++k) sum += rb[k * AW + e];
            out_bias[e] = sum;
        }
    }
}

// order-fixed reduction of the partials: out[o][c] = sum_p partial[p][o][c] for c < cols (padding columns dropped); the
// threads past rows * cols reduce the bias partials.  The additions run in index order p = 0, 1, 2, ... (deterministic);
// the loads of 16 partials are issued together so the loop is bandwidth- rather than latency-bound.  One launch serves a
// batch of jobs (blockIdx.y = job).
struct DwReduceJob { const float* partial; const float* partial_bias; float* out; float* out_bias; int32_t out_ld, out_col0; };
struct DwReduceBatch { DwReduceJob job[DW_MAX_JOBS]; };
__global__ void dw_reduce_batch_kernel(const DwReduceBatch rb, int parts, int rows, int ld, int cols) {
    const DwReduceJob j = rb.job[blockIdx.y];
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t elems = (int64_t)rows * cols;
    const float* p;
    int64_t stride;
    float* dst;
    if (t < elems) {
        const int o = (int)(t / cols), c = (int)(t - (int64_t)o * cols);
        p = j.partial + (int64_t)o * ld + c;
        stride = (int64_t)rows * ld;
        dst = j.out + (int64_t)o * j.out_ld + j.out_col0 + c;
    } else if (j.out_bias && t < elems + rows) {
        const int o = (int)(t - elems);
        p = j.partial_bias + o;
        stride = rows;
        dst = j.out_bias + o;
    } else {
        return;
    }
    float s = 0.0f;
    int k = 0;
    for (; k + 16 <= parts; k += 16) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = p[(k + u) * stride];
#pragma unroll
        for (int u = 0; u < 16; ++u) s += v[u];
    }
    for (; k < parts; ++k) s += p[k * stride];
    *dst = s;
}

// ---- the 4-row heads --------------------------------------------------------------------------------------------
// fc_alpha (1 row) and fc_rgb (3 rows) share the delta dlast (n, 4): out[r][k] = sum_n dlast[n][r] * act[n][k] for
// act = h[L-1] (fc_alpha takes row 3) and act = v (fc_rgb takes rows 0..2).  An MFMA tile would waste 12 of its 16
// rows and the product is HBM-bound anyway (act is read once: 2 FLOP / B), so this is a VALU kernel: a workgroup owns a
// contiguous slice of samples, a thread four columns (one 16-byte load per row; a 256-wide row is one wavefront-wide
// 1 KiB load, dlast[n] a broadcast 16-byte load), 16 fp32 accumulators.  The slice partials go through the same
// order-fixed second pass as the MFMA kernel's (head_reduce_kernel: parts in index order -> deterministic).
template <int K>
__global__ __launch_bounds__(256) void head_grad_kernel(const float* __restrict__ dlast, const float* __restrict__ act,
                                                        int64_t n, int rows_per_part, float* __restrict__ partial,
                                                        float* __restrict__ partial_bias) {
    constexpr int TPR = K / 4, RPP = 256 / TPR;            // threads per row, rows per pass
    __shared__ float4 red[RPP][4][TPR];
    __shared__ float4 bred[RPP];
    const int c4 = threadIdx.x % TPR, rg = threadIdx.x / TPR;
    const int64_t n0 = (int64_t)blockIdx.x * rows_per_part;
    const int64_t n1 = n0 + rows_per_part < n ? n0 + rows_per_part : n;
    const float4* a4 = reinterpret_cast<const float4*>(act);
    const float4* d4 = reinterpret_cast<const float4*>(dlast);
    float4 acc[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 bs = make_float4(0.f, 0.f, 0.f, 0.f);
    auto fma_row = [&](const float4& x, const float4& d) {
        acc[0].x += d.x * x.x; acc[0].y += d.x * x.y; acc[0].z += d.x * x.z; acc[0].w += d.x * x.w;
        acc[1].x += d.y * x.x; acc[1].y += d.y * x.y; acc[1].z += d.y * x.z; acc[1].w += d.y * x.w;
        acc[2].x += d.z * x.x; acc[2].y += d.z * x.y; acc[2].z += d.z * x.z; acc[2].w += d.z * x.w;
        acc[3].x += d.w * x.x; acc[3].y += d.w * x.y; acc[3].z += d.w * x.z; acc[3].w += d.w * x.w;
        bs.x += d.x; bs.y += d.y; bs.z += d.z; bs.w += d.w;
    };
    int64_t i = n0 + rg;
    for (; i + 7 * RPP < n1; i += 8 * RPP) {
        float4 x[8], d[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { x[u] = a4[(i + u * RPP) * TPR + c4]; d[u] = d4[i + u * RPP]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) fma_row(x[u], d[u]);
    }
    for (; i < n1; i += RPP) fma_row(a4[i * TPR + c4], d4[i]);
#pragma unroll
    for (int r = 0; r < 4; ++r) red[rg][r][c4] = acc[r];
    if (c4 == 0) bred[rg] = bs;
    __syncthreads();
    if (threadIdx.x < 4 * TPR) {                           // (row r, column quad c): row groups added in index order
        const int r = threadIdx.x / TPR, c = threadIdx.x % TPR;
        float4 s = red[0][r][c];
        for (int q = 1; q < RPP; ++q) { const float4 v = red[q][r][c]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
        reinterpret_cast<float4*>(partial)[((int64_t)blockIdx.x * 4 + r) * TPR + c] = s;
    }
    if (threadIdx.x == 0) {
        float4 s = bred[0];
        for (int q = 1; q < RPP; ++q) { const float4 v = bred[q]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
        reinterpret_cast<float4*>(partial_bias)[blockIdx.x] = s;
    }
}

// out[e] = sum_p partial[p][e], e < elems (+ the 4 bias sums behind them): 4 part-groups per element added up
// sequentially (16 loads in flight), then the 4 group sums in index order.
__global__ __launch_bounds__(256) void head_reduce_kernel(const float* __restrict__ partial,
                                                          const float* __restrict__ partial_bias, int parts, int elems,
                                                          float* __restrict__ out, float* __restrict__ out_bias) {
    __shared__ float grp[4][64];
    const int e = blockIdx.x * 64 + (threadIdx.x & 63), pg = threadIdx.x >> 6;
    const int per = (parts + 3) / 4;
    const int p0 = pg * per, p1 = p0 + per < parts ? p0 + per : parts;
    const float* p = nullptr;
    int64_t stride = 0;
    if (e < elems) { p = partial + e; stride = elems; }
    else if (e < elems + 4) { p = partial_bias + (e - elems); stride = 4; }
    float s = 0.0f;
    if (p) {
        int k = p0;
        for (; k + 16 <= p1; k += 16) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = p[(k + u) * stride];
#pragma unroll
            for (int u = 0; u < 16; ++u) s += v[u];
        }
        for (; k < p1; ++k) s += p[k * stride];
    }
    grp[pg][threadIdx.x & 63] = s;
    __syncthreads();
    if (pg == 0 && p) {
        const float t = ((grp[0][threadIdx.x] + grp[1][threadIdx.x]) + grp[2][threadIdx.x]) + grp[3][threadIdx.x];
        if (e < elems) out[e] = t;
        else if (out_bias) out_bias[e - elems] = t;
    }
}

constexpr int HEAD_MAX_PARTS = 512;

struct DwPlan {
    int ab, bb, ksplit, rows;
    void (*kernel)(const DwBatch);
};
// per shape: the 16-row kernel (any n % 16 == 0) and, for the products narrower than 256 x 256, the 32-row one (n % 32 == 0),
// listed first so that it is preferred
static const DwPlan g_dw_plans[] = {
    {4, 4, 1, 16, &dw_kernel<4, 4>},
    {4, 1, 2, 16, &dw_kernel<4, 1>},
    {2, 4, 1, 16, &dw_kernel<2, 4>},
    {2, 1, 4, 32, &dw_kernel<2, 1, 32>}, {2, 1, 4, 16, &dw_kernel<2, 1>},
    {2, 2, 2, 32, &dw_kernel<2, 2, 32>}, {2, 2, 2, 16, &dw_kernel<2, 2>},
    {1, 2, 4, 32, &dw_kernel<1, 2, 32>}, {1, 2, 4, 16, &dw_kernel<1, 2>},
    // 64 x 64 (the hidden layers of the 64-wide networks, BASELINE config 1): ONE 64 x 64 block, every wave the whole block on an
    // eighth of a chunk's k-groups -- HBM-bound (16 FLOP / B), so what counts is rows in flight: 64-row chunks
    {1, 1, 8, 64, &dw_kernel<1, 1, 64>}, {1, 1, 8, 32, &dw_kernel<1, 1, 32>},
};

static const DwPlan* pick_dw_plan(int ab, int bb, int64_t n) {
    static const int force = getenv("NM_DW_ROWS") ? atoi(getenv("NM_DW_ROWS")) : 0;      // A/B hook of the tools: 16 | 32
    for (const DwPlan& p : g_dw_plans)
        if (p.ab == ab && p.bb == bb && n % p.rows == 0 && (!force || p.rows == force)) return &p;
    return nullptr;
}

}  // namespace nm

using namespace nm;

extern "C" int64_t nm_weight_grad_workspace_bytes(int32_t out_features, int32_t act_stride, int32_t num_cus) {
    return nm_weight_grad_workspace_bytes_ex(out_features, out_features, act_stride, act_stride, num_cus);
}

namespace nm {

int64_t weight_grad_tuned_workspace_bytes(int32_t out_features, int32_t act_stride, int32_t num_cus) {
    if (out_features % 64 || act_stride % 64) return 0;
    for (const DwPlan& p : g_dw_plans)
        if (p.ab == out_features / 64 && p.bb == act_stride / 64)
            return (int64_t)num_cus * p.ksplit * ((int64_t)out_features * act_stride + out_features) * 4;
    return 0;
}

// The kernel above for the shipped configs' six (out, stride) pairs, n a multiple of 16, delta rows of stride out_features:
// `jobs` products of that one shape and row count in one launch.  Returns -1 when it does not serve the call
// (nm_weight_grad_ex / nm_weight_grad_batch then take the general kernel, nerf_dw_g.hip).
int weight_grad_tuned(int device_cus, int jobs, const nm_weight_grad_job* job, int32_t out_features, int32_t act_stride,
                      int32_t in_features, int64_t n, void* d_workspace, hipStream_t stream) {
    if (n % DW_ROWS || out_features % 64 || act_stride % 64 || jobs < 1 || jobs > DW_MAX_JOBS) return -1;
    const int ab = out_features / 64, bb = act_stride / 64;
    const DwPlan* plan = pick_dw_plan(ab, bb, n);
    if (!plan) return -1;
    const int64_t chunks = n / plan->rows;
    const int cus = device_cus > 0 ? device_cus : 256;
    int64_t per_job = cus / jobs;
    per_job = per_job < 1 ? 1 : (per_job > chunks ? chunks : per_job);
    const int parts = (int)per_job;          // one partial per workgroup: the k-split waves add up in LDS (dw_kernel's epilogue)
    const int64_t job_floats = (int64_t)parts * ((int64_t)out_features * act_stride + out_features);
    DwBatch batch;
    DwReduceBatch rb;
    batch.jobs = jobs; batch.per_job = (int)per_job;
    for (int j = 0; j < jobs; ++j) {
        DwArgs& a = batch.job[j];
        a.a = job[j].d_delta; a.b = job[j].d_act; a.chunks = chunks;
        a.partial = static_cast<float*>(d_workspace) + j * job_floats;
        a.partial_bias = a.partial + (int64_t)parts * out_features * act_stride;
        rb.job[j] = DwReduceJob{a.partial, a.partial_bias, job[j].d_dw, job[j].d_dbias, job[j].dw_ld, job[j].dw_col0};
    }
    const int ring_bytes = 4 * plan->rows * (out_features + act_stride) * 4;
    const int fold_bytes = plan->ksplit > 1 ? plan->ksplit * (out_features * act_stride + out_features) * 4 : 0;   // the epilogue's k-part sums
    const int lds_bytes = ring_bytes > fold_bytes ? ring_bytes : fold_bytes;
    if (int rc = ensure_dynamic_lds((const void*)plan->kernel, lds_bytes)) return rc;
    hipLaunchKernelGGL(plan->kernel, dim3((unsigned)(per_job * jobs)), dim3(512), lds_bytes, stream, batch);
    const int64_t elems = (int64_t)out_features * in_features;
    hipLaunchKernelGGL(dw_reduce_batch_kernel, dim3((unsigned)((elems + out_features + 255) / 256), jobs), dim3(256), 0, stream, rb, parts,
                       out_features, act_stride, in_features);
    NM_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace nm

// d_delta (n, out_features) and d_act (n, act_stride) row-major.  Writes d_dw[o * dw_ld + dw_col0 + c] for c < in_features
// and, when d_dbias != NULL, d_dbias[o] = sum_n delta[n][o].  = nm_weight_grad_ex with delta rows of stride out_features.
extern "C" int nm_weight_grad(int device_cus, const float* d_delta, int32_t out_features, const float* d_act,
                              int32_t act_stride, int32_t in_features, int64_t n, void* d_workspace, float* d_dw,
                              int32_t dw_ld, int32_t dw_col0, float* d_dbias, void* stream_) {
    return nm_weight_grad_ex(device_cus, d_delta, out_features, out_features, d_act, in_features, act_stride, n, d_workspace,
                             d_dw, dw_ld, dw_col0, d_dbias, stream_);
}

extern "C" int64_t nm_head_grad_workspace_bytes(int32_t in_features) { return nm_head_grad_workspace_bytes_ex(in_features); }

namespace nm {

// head_grad_kernel<K> for contiguous 64- / 128- / 256-wide rows; -1 when it does not serve the call (-> nerf_dw_g.hip)
int head_grad_tuned(const float* d_dlast, const float* d_act, int32_t in_features, int64_t n, void* d_workspace, float* d_dw,
                    float* d_dbias, hipStream_t stream) {
    if (!(in_features == 64 || in_features == 128 || in_features == 256)) return -1;
    const int rpp = 256 / (in_features / 4);
    int64_t rows = (n + HEAD_MAX_PARTS - 1) / HEAD_MAX_PARTS;
    rows = (rows + 8 * rpp - 1) / (8 * rpp) * (8 * rpp);        // whole unrolled passes
    const int parts = (int)((n + rows - 1) / rows);
    float* partial = static_cast<float*>(d_workspace);
    float* partial_bias = partial + (int64_t)HEAD_MAX_PARTS * 4 * in_features;
    if (in_features == 256)
        hipLaunchKernelGGL(head_grad_kernel<256>, dim3(parts), dim3(256), 0, stream, d_dlast, d_act, n, (int)rows, partial, partial_bias);
    else if (in_features == 128)
        hipLaunchKernelGGL(head_grad_kernel<128>, dim3(parts), dim3(256), 0, stream, d_dlast, d_act, n, (int)rows, partial, partial_bias);
    else
        hipLaunchKernelGGL(head_grad_kernel<64>, dim3(parts), dim3(256), 0, stream, d_dlast, d_act, n, (int)rows, partial, partial_bias);
    const int elems = 4 * in_features;
    hipLaunchKernelGGL(head_reduce_kernel, dim3((elems + 4 + 63) / 64), dim3(256), 0, stream, partial, partial_bias, parts,
                       elems, d_dw, d_dbias);
    NM_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace nm

// d_dlast (n, 4) and d_act (n, in_features) row-major.  Writes d_dw[r * in_features + k] = sum_n dlast[n][r] * act[n][k]
// (4 rows) and, when d_dbias != NULL, d_dbias[r] = sum_n dlast[n][r].  = nm_head_grad_ex with rows of stride in_features.
extern "C" int nm_head_grad(const float* d_dlast, const float* d_act, int32_t in_features, int64_t n, void* d_workspace,
                            float* d_dw, float* d_dbias, void* stream_) {
    return nm_head_grad_ex(d_dlast, d_act, in_features, in_features, n, d_workspace, d_dw, d_dbias, stream_);
}
