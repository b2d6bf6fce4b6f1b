// Weight gradients for EVERY layer shape and every sample count (gfx950): dW[out][in] = sum_n delta[n][out] * act[n][in] and
// db[out] = sum_n delta[n][out] for row-major operands of any width, any row stride and any n -- what autograd's addmm
// backward computes for each Linear of /root/reference/src/nerf/models.py:5-58,60-80, whatever hidden_size /
// num_encoding_fn_* the constructor was given (and for a ragged last batch of /root/reference/src/models/model_nerf.py:88-151).
// nerf_dw.hip holds the kernel tuned for the six (out, stride) pairs of the shipped configs (64-feature blocks, b128 operand
// reads, n % 16 == 0); until round 5 every other product -- 64-wide networks, the generic-shape family, n % 16 != 0 -- went
// to a library GEMM.  This is the same dataflow with its compile-time geometry made run-time:
//
//   * one workgroup (8 waves) per CU owns a contiguous range of samples and a block of <= (WA*TA) x (WB*TB) 16 x 16 output
//     tiles, accumulated in registers over the whole range (v_mfma_f32_16x16x4_f32, the instruction's contraction index is
//     the sample index); partials are summed by a second, ORDER-FIXED pass (deterministic, no floating-point atomics).
//     TA x TB (tiles per wave) is the template; how the 8 waves are arranged -- WA x WB over the output block, WK ways over
//     the samples of a chunk -- and how many blocks the output is cut into (grid y / z) are run-time values chosen by the
//     host planner below from the shape alone, so that small products still occupy 8 waves and 400-wide ones fit.
//   * the operand rows are DMA'd HBM -> LDS as they lie in memory (scalar base, buffer_load ... lds), three chunks ahead
//     into a 4-slot ring, with a counted s_waitcnt in front of one bare s_barrier per chunk (nerf_dw.hip's schedule).  A
//     chunk of `rows` rows of an unsplit operand is ONE contiguous run of rows * ld floats whatever ld is; a column block of
//     a split operand is one run per row.  16-byte pieces when base and stride allow it, 4-byte pieces otherwise (a 50-wide
//     view layer, a 63-wide encoding).
//   * NO PADDING IS EVER WRITTEN OR REQUIRED.  Columns: an MFMA output element depends on one row of A and one column of B
//     only, so whatever a tile reads beyond the real width (the next row's values, stale LDS) lands in accumulators of
//     dW entries >= out / >= in that the reduction never reads.  Rows: the buffer descriptor's num_records is the exact end
//     of the operand, so the rows of the last chunk beyond n arrive as zeros (out-of-range buffer loads return 0, also into LDS).
//     The exact-extent (raw, VGPR-addressed) descriptor serves the LAST chunk only; interior chunks go through the
//     ADD_TID form, which reads no address VGPR -- and makes no range check at all (probed: tests/tools/probes/).
//   * operands come out of LDS with ds_read_b32 at (row, 16 t + i): correct whatever the stride's alignment (at most 2-way
//     bank conflicts); a wave reads TA + TB values per TA * TB MFMAs.  The 4 x 8 / 4 x 4 geometries also exist with the
//     tuned kernel's ds_read_b128 mapping (VEC) for 16-byte-aligned LDS rows.
//   * several products of one shape run as ONE launch (DwGBatch: workgroup x -> job x / per_job): the L same-shape layers
//     of a network share the CUs, each with 1 / L of the partials (nm_weight_grad_batch).
//   * the same kernel is the GEMM of the layer-wise network path (dwg_gemm: a short contraction, a wide output, one sample
//     part, bias / activation / mask applied to the accumulators on the way out: nerf_layerwise.hip).
//
// Roofline: MFMA for wide layers (2 * out * in FLOP per sample against 4 * (out + in) bytes), HBM for narrow ones (64 x 64:
// 16 FLOP/B).
#include <cstdlib>
#include <map>
#include <mutex>
#include <utility>

#include "nm_internal.h"
#include "mlp_device.h"

namespace nm {

struct DwOperand {
    const float* base;       // row 0 of the operand
    int64_t bytes;           // n * ld * 4: the operand's exact extent
    int32_t ld;              // floats per row in HBM
    int32_t ls;              // floats per row in LDS
    int32_t x4;              // 16-byte (1) or 4-byte (0) DMA elements: 1 KiB / 256 B per wave instruction
    int32_t nseg;            // contiguous runs per chunk image: 1 (whole rows) or `rows` (a column block)
    int32_t pps;             // DMA instructions per run
    int32_t gstride;         // bytes between runs in HBM
    int32_t lstride;         // bytes between runs in LDS
    int32_t blk_cols;        // columns the block index of this operand advances by (0: not split)
    int32_t img;             // LDS bytes of a chunk image
};

struct DwGArgs {
    DwOperand A, B;          // delta rows (n, out), activation rows (n, in)
    int64_t n;
    int32_t rows;            // rows per chunk: a multiple of 4 * wk
    int32_t wa, wb, wk;      // wave arrangement, wa * wb * wk == 8
    int32_t out_pad, in_pad; // dimensions of a partial: grid.y * wa * TA * 16, grid.z * wb * TB * 16
    float* partial;          // (gridDim.x * wk, out_pad, in_pad)
    float* partial_bias;     // (gridDim.x * wk, out_pad)
    // optional fused epilogue (dwg_gemm, one sample part): instead of the partial, out[row][col] = act(acc + bias[row]) (masked
    // by mask[row][col] > 0) straight from the accumulators -- the layer-wise path's bias / ReLU / sigmoid / ReLU' pass
    float* epi_out;          // (>= epi_rows x epi_ld) plane, or null
    const float* epi_bias;   // per output row, or null
    const float* epi_mask;   // plane of leading dimension epi_ld, or null
    int64_t epi_ld;
    int32_t epi_rows, epi_act;     // rows of the real output; 0 none, 1 ReLU, 2 sigmoid
};

// a batch of products of ONE shape, stride set and row count (nm_weight_grad_batch): everything in `common` but the four
// per-job pointers.  Workgroup x serves job x / per_job, sample part x % per_job (see DwBatch in nerf_dw.hip for the why).
constexpr int DWG_MAX_JOBS = 16;
struct DwGJob { const float* a; const float* b; float* partial; float* partial_bias; };
struct DwGBatch {
    DwGArgs common;
    DwGJob job[DWG_MAX_JOBS];
    int32_t jobs, per_job;
};

#define NM_VMCNT_CASE(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
__device__ __forceinline__ void wait_vmcnt(int pieces) {     // wave-uniform run-time count (the instruction takes an immediate)
    switch (pieces) {
        NM_VMCNT_CASE(1) NM_VMCNT_CASE(2) NM_VMCNT_CASE(3) NM_VMCNT_CASE(4) NM_VMCNT_CASE(5) NM_VMCNT_CASE(6) NM_VMCNT_CASE(7)
        NM_VMCNT_CASE(8) NM_VMCNT_CASE(9) NM_VMCNT_CASE(10) NM_VMCNT_CASE(11) NM_VMCNT_CASE(12) NM_VMCNT_CASE(13)
        NM_VMCNT_CASE(14) NM_VMCNT_CASE(15) NM_VMCNT_CASE(16) NM_VMCNT_CASE(17) NM_VMCNT_CASE(18) NM_VMCNT_CASE(19)
        NM_VMCNT_CASE(20) NM_VMCNT_CASE(21) NM_VMCNT_CASE(22) NM_VMCNT_CASE(23) NM_VMCNT_CASE(24) NM_VMCNT_CASE(25)
        NM_VMCNT_CASE(26) NM_VMCNT_CASE(27) NM_VMCNT_CASE(28) NM_VMCNT_CASE(29) NM_VMCNT_CASE(30) NM_VMCNT_CASE(31)
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}
#undef NM_VMCNT_CASE

// This wave's share of one operand's DMA: pieces wave, wave + 8, ... of the nseg * pps pieces of a chunk image.  Where a
// piece comes from and goes to is computed ONCE, piece p of the wave in lane p of two registers; the per-chunk loop fetches
// them with v_readlane (the first version walked (run, piece-in-run) counters in SGPRs: 45 scalar instructions per k-group
// against the tuned kernel's 13, and the matrix pipe 74 % busy against 89 %).
struct DwStream {
    __amdgpu_buffer_rsrc_t fast;   // ADD_TID_ENABLE, stride = the DMA element: the hardware adds lane * stride, NO address VGPR
                                   // is read (a VGPR-addressed piece holds up the SIMD's MFMA issue for 60 - 180 cycles,
                                   // mlp_device.h) -- and NO range check is made in this mode (probed: tests/tools/probes/)
    __amdgpu_buffer_rsrc_t safe;   // raw descriptor with the exact extent: the LAST chunk only (zeros past the operand)
    int32_t count;           // pieces this wave issues per chunk (<= 64)
    uint32_t goff, loff;     // lane p: byte offset of its piece p inside a chunk in HBM / inside the LDS image
};

// VEC (TA, TB multiples of 4; both LDS row strides multiples of 4 floats): a lane fetches 4 consecutive features of a
// 64-feature block with ONE ds_read_b128 -- register q is then the operand of "tile q" whose index i stands for feature 4 i + q,
// the tuned kernel's mapping: 3 reads instead of 12 per 32 MFMAs, conflict-free for 64-float-multiple strides.
template <int TA, int TB, bool VEC = false>
__global__ __launch_bounds__(512, 2) void dw_kernel_g(const DwGBatch batch) {
    static_assert(!VEC || (TA % 4 == 0 && TB % 4 == 0), "VEC: whole 64-feature blocks per wave");
    const DwGArgs& args = batch.common;
    const int job_i = blockIdx.x / batch.per_job;
    const int part_i = blockIdx.x - job_i * batch.per_job;
    const DwGJob job = batch.job[job_i];
    constexpr int NW = 8;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, i = lane & 15;
    const int wab = args.wa * args.wb;
    const int wk_i = wave / wab, rem = wave - wk_i * wab;
    const int wa_i = rem / args.wb, wb_i = rem - wa_i * args.wb;
    const int rows = args.rows;
    const int64_t chunks = (args.n + rows - 1) / rows;
    const int64_t c_lo = chunks * part_i / batch.per_job, c_hi = chunks * (part_i + 1) / batch.per_job;
    const int slot_bytes = args.A.img + args.B.img;

    // descriptors rebased to this workgroup's first row and column block: offsets stay below 2^32, num_records is what is
    // left of the operand from there (the tail rows of the last chunk and everything past the operand read as zeros)
    auto open = [&](const DwOperand& op, const float* base_ptr, int blk) {
        DwStream s;
        const int64_t skip = (c_lo * rows * op.ld + (int64_t)blk * op.blk_cols) * 4;
        int64_t left = op.bytes - skip;
        left = left < 0 ? 0 : (left > 0xfffffff0ll ? 0xfffffff0ll : left);
        char* base = const_cast<char*>(reinterpret_cast<const char*>(base_ptr) + skip);
        s.safe = __builtin_amdgcn_make_buffer_rsrc(base, (short)0, (int)(uint32_t)left, 0x00020000);
        s.fast = __builtin_amdgcn_make_buffer_rsrc(base, (short)(op.x4 ? 16 : 4), 0x7fffffff, 1 << 23);
        const int total = op.nseg * op.pps;
        s.count = total > wave ? (total - wave + NW - 1) / NW : 0;
        const int u = wave + NW * lane, seg = u / op.pps, k = u - seg * op.pps, unit = op.x4 ? 1024 : 256;
        s.goff = (uint32_t)seg * (uint32_t)op.gstride + (uint32_t)(k * unit);
        s.loff = (uint32_t)seg * (uint32_t)op.lstride + (uint32_t)(k * unit);
        return s;
    };
    const DwStream sa = open(args.A, job.a, blockIdx.y), sb = open(args.B, job.b, blockIdx.z);
    const int my_pieces = sa.count + sb.count;
    const uint32_t lane16 = lane * 16, lane4 = lane * 4;

    // Chunks that lie wholly inside the operand go through the scalar-addressed descriptor; the last chunk of the operand --
    // rows beyond n, and the pieces' rounding past the last row -- through the range-checked one (whole offset in the VGPR:
    // that is the offset the check is defined on).
    auto dma_op = [&](const DwOperand& op, const DwStream& s, int64_t c, char* img) {
        const uint32_t chunk_off = (uint32_t)((c - c_lo) * rows * op.ld * 4);
        if (c + 1 < chunks) {
            if (op.x4) {
                for (int p = 0; p < s.count; ++p) {
                    const uint32_t go = __builtin_amdgcn_readlane(s.goff, p), lo = __builtin_amdgcn_readlane(s.loff, p);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(s.fast, (__attribute__((address_space(3))) void*)(img + lo), 16, 0,
                                                             chunk_off + go, 0, 0);
                }
            } else {
                for (int p = 0; p < s.count; ++p) {
                    const uint32_t go = __builtin_amdgcn_readlane(s.goff, p), lo = __builtin_amdgcn_readlane(s.loff, p);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(s.fast, (__attribute__((address_space(3))) void*)(img + lo), 4, 0,
                                                             chunk_off + go, 0, 0);
                }
            }
        } else if (op.x4) {
            for (int p = 0; p < s.count; ++p) {
                const uint32_t go = __builtin_amdgcn_readlane(s.goff, p), lo = __builtin_amdgcn_readlane(s.loff, p);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(s.safe, (__attribute__((address_space(3))) void*)(img + lo), 16,
                                                         chunk_off + go + lane16, 0, 0, 0);
            }
        } else {
            for (int p = 0; p < s.count; ++p) {
                const uint32_t go = __builtin_amdgcn_readlane(s.goff, p), lo = __builtin_amdgcn_readlane(s.loff, p);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(s.safe, (__attribute__((address_space(3))) void*)(img + lo), 4,
                                                         chunk_off + go + lane4, 0, 0, 0);
            }
        }
    };
    auto dma = [&](int64_t c, int slot) {
        if (c >= c_hi) return;
        char* dst = lds + slot * slot_bytes;
        dma_op(args.A, sa, c, dst);
        dma_op(args.B, sb, c, dst + args.A.img);
    };

    f32x4 acc[TA][TB];
#pragma unroll
    for (int qa = 0; qa < TA; ++qa)
#pragma unroll
        for (int qb = 0; qb < TB; ++qb) acc[qa][qb] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    float bias[TA];
#pragma unroll
    for (int qa = 0; qa < TA; ++qa) bias[qa] = 0.0f;
    const bool bias_owner = wb_i == 0 && blockIdx.z == 0 && job.partial_bias != nullptr;

    // this lane's operands of k-group kg of a chunk: row 4 * (kg * wk + wk_i) + g, columns 16 * (tile) + i
    const int kgw = rows / (4 * args.wk);
    const int step_a = 4 * args.wk * args.A.ls * 4, step_b = 4 * args.wk * args.B.ls * 4;
    const int lane_col = VEC ? 4 * i : i;
    const int lane_a = ((4 * wk_i + g) * args.A.ls + wa_i * TA * 16 + lane_col) * 4;
    const int lane_b = args.A.img + ((4 * wk_i + g) * args.B.ls + wb_i * TB * 16 + lane_col) * 4;
    auto load_ops = [&](int slot, int kg, float (&a)[TA], float (&b)[TB]) {
        const char* pa = lds + slot * slot_bytes + kg * step_a + lane_a;
        const char* pb = lds + slot * slot_bytes + kg * step_b + lane_b;
        if constexpr (VEC) {
#pragma unroll
            for (int blk = 0; blk < TA / 4; ++blk) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(pa + blk * 256);
#pragma unroll
                for (int q = 0; q < 4; ++q) a[4 * blk + q] = v[q];
            }
#pragma unroll
            for (int blk = 0; blk < TB / 4; ++blk) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(pb + blk * 256);
#pragma unroll
                for (int q = 0; q < 4; ++q) b[4 * blk + q] = v[q];
            }
        } else {
#pragma unroll
            for (int qa = 0; qa < TA; ++qa) a[qa] = *reinterpret_cast<const float*>(pa + qa * 64);
#pragma unroll
            for (int qb = 0; qb < TB; ++qb) b[qb] = *reinterpret_cast<const float*>(pb + qb * 64);
        }
    };
    // one k-group: the operands of the NEXT one are put in flight first (pinned: hipcc otherwise sinks the reads to their use)
    auto kstep = [&](const float (&a)[TA], const float (&b)[TB], float (&an)[TA], float (&bn)[TB], int nslot, int nkg) {
        load_ops(nslot, nkg, an, bn);
        __builtin_amdgcn_sched_barrier(0);
        if (bias_owner) {                                 // column sums of delta: a real (wave-uniform) branch, not selects
            asm volatile("" ::: );
#pragma unroll
            for (int qa = 0; qa < TA; ++qa) bias[qa] += a[qa];
        }
#pragma unroll
        for (int qa = 0; qa < TA; ++qa)
#pragma unroll
            for (int qb = 0; qb < TB; ++qb)
                acc[qa][qb] = VEC ? __builtin_amdgcn_mfma_f32_16x16x4f32(a[qa], b[qb], acc[qa][qb], 0, 0, 0)
                                  : __builtin_amdgcn_mfma_f32_16x16x4f32(b[qb], a[qa], acc[qa][qb], 0, 0, 0);
    };

    dma(c_lo, 0);
    dma(c_lo + 1, 1);
    dma(c_lo + 2, 2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float a0[TA], b0[TB], a1[TA], b1[TB];
    load_ops(0, 0, a0, b0);
    int slot = 0;
    // the two waves of a SIMD (w, w + 4) issue their DMA half a chunk apart
    const int dma_kg = wave >= NW / 2 ? (kgw / 2) & ~1 : 0;
    for (int64_t c = c_lo; c < c_hi; ++c) {
        const int slot1 = (slot + 1) & 3;
        int kg = 0;
        for (; kg + 2 <= kgw; kg += 2) {                  // two k-groups per trip: the operand registers ping-pong
            if (kg == dma_kg) dma(c + 3, (slot + 3) & 3);
            kstep(a0, b0, a1, b1, slot, kg + 1);
            const bool last = kg + 2 == kgw;               // next chunk: visible since the last barrier
            kstep(a1, b1, a0, b0, last ? slot1 : slot, last ? 0 : kg + 2);
        }
        if (kgw & 1) {
            if (kg == dma_kg) dma(c + 3, (slot + 3) & 3);
            kstep(a0, b0, a1, b1, slot1, 0);
#pragma unroll
            for (int qa = 0; qa < TA; ++qa) a0[qa] = a1[qa];
#pragma unroll
            for (int qb = 0; qb < TB; ++qb) b0[qb] = b1[qb];
        }
        // chunk c + 2 must have landed before anybody reads it (from the end of chunk c + 1 on); chunk c + 3, issued during
        // this iteration, may stay in flight across the barrier
        if (c + 3 < c_hi) wait_vmcnt(my_pieces);
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        slot = slot1;
    }

    // ---- this workgroup's partial.  The MFMA's row operand is the ACTIVATION tile, its column operand the delta tile, so
    //      tile (qa, qb) register r of lane (g, i) is dW[16 ta + i][16 tb + 4 g + r]: four consecutive floats of a row of dW
    //      per lane, one 16-byte store per tile
    const int64_t part = (int64_t)part_i * args.wk + wk_i;
    const int row0 = (blockIdx.y * args.wa + wa_i) * TA * 16, col0 = (blockIdx.z * args.wb + wb_i) * TB * 16;
    // element (row, 4 consecutive columns) of the output a lane holds for (qa, group of 4 registers / tiles):
    //   plain: tile (qa, qb), registers r = 0..3  -> row 16 qa + i,          columns 16 qb + 4 g + (0..3)
    //   VEC:   tiles (qa, 4 blk + 0..3), register r -> row 16 (qa / 4 * 4) + 4 (4 g + r) + qa % 4,  columns 64 blk + 4 i + (0..3)
    auto emit = [&](auto&& sink) {
        if constexpr (VEC) {
#pragma unroll
            for (int qa = 0; qa < TA; ++qa)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = row0 + 64 * (qa / 4) + 4 * (4 * g + r) + (qa % 4);
#pragma unroll
                    for (int blk = 0; blk < TB / 4; ++blk) {
                        const f32x4 v = {acc[qa][4 * blk][r], acc[qa][4 * blk + 1][r], acc[qa][4 * blk + 2][r], acc[qa][4 * blk + 3][r]};
                        sink(row, col0 + 64 * blk + 4 * i, v);
                    }
                }
        } else {
#pragma unroll
            for (int qa = 0; qa < TA; ++qa)
#pragma unroll
                for (int qb = 0; qb < TB; ++qb) sink(row0 + 16 * qa + i, col0 + 16 * qb + 4 * g, acc[qa][qb]);
        }
    };
    if (args.epi_out) {
        emit([&](int row, int col, f32x4 v) {
            if (row >= args.epi_rows) return;
            const float bias_r = args.epi_bias ? args.epi_bias[row] : 0.0f;
            const int64_t at = (int64_t)row * args.epi_ld + col;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float x = v[r] + bias_r;
                if (args.epi_act == 1) x = fmaxf(x, 0.0f);
                else if (args.epi_act == 2) x = 1.0f / (1.0f + expf(-x));
                v[r] = x;
            }
            if (args.epi_mask) {
                const f32x4 mk = *reinterpret_cast<const f32x4*>(args.epi_mask + at);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = mk[r] > 0.0f ? v[r] : 0.0f;
            }
            *reinterpret_cast<f32x4*>(args.epi_out + at) = v;
        });
        return;
    }
    float* out = job.partial + part * ((int64_t)args.out_pad * args.in_pad);
    emit([&](int row, int col, f32x4 v) { *reinterpret_cast<f32x4*>(out + (int64_t)row * args.in_pad + col) = v; });
    if (bias_owner) {
        // lane (g, i) holds the sum over its rows (= g mod 4 of its k-groups) of one feature per qa: fold the 4 lane groups
#pragma unroll
        for (int qa = 0; qa < TA; ++qa) {
            float v = bias[qa];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            const int feature = VEC ? 64 * (qa / 4) + 4 * i + (qa % 4) : 16 * qa + i;
            if (g == 0) job.partial_bias[part * args.out_pad + row0 + feature] = v;
        }
    }
}

// order-fixed reduction of the partials (parts in index order, 16 loads in flight); threads past rows * cols do the biases;
// blockIdx.y = job of a batch
struct DwGReduceJob { const float* partial; const float* partial_bias; float* out; float* out_bias; int32_t out_ld, out_col0; };
struct DwGReduceBatch { DwGReduceJob job[DWG_MAX_JOBS]; };
__global__ void dw_reduce_g_kernel(const DwGReduceBatch rb, int parts, int64_t part_stride, int bias_stride, int rows, int ld, int cols) {
    const DwGReduceJob j = rb.job[blockIdx.y];
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t elems = (int64_t)rows * cols;
    const float* p;
    int64_t stride;
    float* dst;
    if (t < elems) {
        const int o = (int)(t / cols), c = (int)(t - (int64_t)o * cols);
        p = j.partial + (int64_t)o * ld + c;
        stride = part_stride;
        dst = j.out + (int64_t)o * j.out_ld + j.out_col0 + c;
    } else if (j.out_bias && t < elems + rows) {
        const int o = (int)(t - elems);
        p = j.partial_bias + o;
        stride = bias_stride;
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

// ---- the 4-row heads for any activation width ---------------------------------------------------------------------------
// out[r][k] = sum_n dlast[n][r] * act[n][k] (fc_alpha / fc_rgb / fc_out share the 4-wide delta): HBM-bound VALU kernel as
// head_grad_kernel<K> of nerf_dw.hip, with the row width and stride run-time values.  A thread owns VEC columns of a group
// of rows; the workgroup's row groups are added up in index order, the workgroups' partials by head_reduce_kernel's scheme.
template <int VEC>
__global__ __launch_bounds__(256) void head_grad_g_kernel(const float* __restrict__ dlast, const float* __restrict__ act,
                                                          int ld, int K, int tpr, int64_t n, int rows_per_part,
                                                          float* __restrict__ partial, float* __restrict__ partial_bias) {
    extern __shared__ float red[];                          // [rpp][4][tpr * VEC] + [rpp][4]
    const int rpp = 256 / tpr;
    const int c = threadIdx.x % tpr, rg = threadIdx.x / tpr;
    const int col = (blockIdx.y * tpr + c) * VEC;
    const bool live = rg < rpp && col < K;
    const int64_t n0 = (int64_t)blockIdx.x * rows_per_part;
    const int64_t n1 = n0 + rows_per_part < n ? n0 + rows_per_part : n;
    float acc[4][VEC];
    float bs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[r][v] = 0.f;
    if (live) {
        auto fma_row = [&](const float (&x)[VEC], const float4& d) {
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                acc[0][v] += d.x * x[v]; acc[1][v] += d.y * x[v]; acc[2][v] += d.z * x[v]; acc[3][v] += d.w * x[v];
            }
            bs[0] += d.x; bs[1] += d.y; bs[2] += d.z; bs[3] += d.w;
        };
        auto load_row = [&](int64_t row, float (&x)[VEC]) {
            const float* p = act + row * ld + col;
            if constexpr (VEC == 4) {
                const float4 v = *reinterpret_cast<const float4*>(p);
                x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w;
            } else {
                x[0] = *p;
            }
        };
        const float4* d4 = reinterpret_cast<const float4*>(dlast);
        int64_t i = n0 + rg;
        for (; i + 7 * (int64_t)rpp < n1; i += 8 * rpp) {
            float x[8][VEC];
            float4 d[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { load_row(i + u * rpp, x[u]); d[u] = d4[i + u * rpp]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) fma_row(x[u], d[u]);
        }
        for (; i < n1; i += rpp) {
            float x[VEC];
            load_row(i, x);
            fma_row(x, d4[i]);
        }
    }
    const int w = tpr * VEC;
    float* bred = red + rpp * 4 * w;
    if (rg < rpp) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int v = 0; v < VEC; ++v) red[(rg * 4 + r) * w + c * VEC + v] = acc[r][v];
        if (c == 0)
#pragma unroll
            for (int r = 0; r < 4; ++r) bred[rg * 4 + r] = bs[r];
    }
    __syncthreads();
    // (row r, column k of this column group): row groups added in index order
    for (int e = threadIdx.x; e < 4 * w; e += 256) {
        const int r = e / w, k = e - r * w;
        const int kk = blockIdx.y * w + k;
        if (kk >= K) continue;
        float s = red[r * w + k];
        for (int q = 1; q < rpp; ++q) s += red[(q * 4 + r) * w + k];
        partial[((int64_t)blockIdx.x * 4 + r) * K + kk] = s;
    }
    if (blockIdx.y == 0 && threadIdx.x < 4) {
        float s = bred[threadIdx.x];
        for (int q = 1; q < rpp; ++q) s += bred[q * 4 + threadIdx.x];
        partial_bias[blockIdx.x * 4 + threadIdx.x] = s;
    }
}

__global__ __launch_bounds__(256) void head_reduce_g_kernel(const float* __restrict__ partial,
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

// ---- host side: the planner ------------------------------------------------------------------------------------------
typedef void (*DwGKernel)(const DwGBatch);
constexpr int DWG_MAX_TA = 6, DWG_MAX_TB = 8, DWG_MAX_TILES = 36;     // tiles per wave: TA x TB <= 36 (144 accumulator registers)
template <int TA, int TB>
constexpr DwGKernel dwg_kernel_or_null() {
    if constexpr (TA * TB <= DWG_MAX_TILES) return &dw_kernel_g<TA, TB>;
    else return nullptr;
}
// the ds_read_b128 variants (whole 64-feature blocks per wave): picked when both operands' LDS rows are 16-byte aligned
static DwGKernel dwg_vec_kernel(int ta, int tb) {
    if (ta == 4 && tb == 8) return &dw_kernel_g<4, 8, true>;
    if (ta == 4 && tb == 4) return &dw_kernel_g<4, 4, true>;
    return nullptr;
}
static bool dwg_vec_ok(const DwGArgs& a) { return a.A.x4 && a.B.x4 && a.A.ls % 4 == 0 && a.B.ls % 4 == 0 && !getenv("NM_DW_NO_VEC"); }
#define NM_DWG_ROW(TA)                                                                                                          \
    { dwg_kernel_or_null<TA, 1>(), dwg_kernel_or_null<TA, 2>(), dwg_kernel_or_null<TA, 3>(), dwg_kernel_or_null<TA, 4>(),         \
      dwg_kernel_or_null<TA, 5>(), dwg_kernel_or_null<TA, 6>(), dwg_kernel_or_null<TA, 7>(), dwg_kernel_or_null<TA, 8>() }
static const DwGKernel g_dwg_kernels[DWG_MAX_TA][DWG_MAX_TB] = {NM_DWG_ROW(1), NM_DWG_ROW(2), NM_DWG_ROW(3), NM_DWG_ROW(4), NM_DWG_ROW(5),
                                                                NM_DWG_ROW(6)};
#undef NM_DWG_ROW

constexpr int DWG_LDS_BYTES = 160 * 1024;
constexpr int DWG_SLACK = 2048;          // what the last image's padded-tile reads may run past it
constexpr int HEAD_G_MAX_PARTS = 512;

struct DwGPlan {
    int nba, nbb, wa, wb, wk, ta, tb, rows, grid_x;
    int lsa, lsb;            // LDS row strides (floats) -- DMA element size NOT included (it only changes the image rounding)
    double cost;
};

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// DMA element of an operand: 16 bytes when base and stride allow -- except for a NARROW column block of a split operand, whose
// rows round up to whole pieces in LDS: an 80-float block is 1 KiB per row with 16-byte pieces, 512 B with 4-byte ones (what
// lets a 400-wide product take five 80-row blocks inside the LDS budget)
static inline bool pick_x4(bool aligned, bool split, int blk_cols) {
    if (!aligned) return false;
    if (!split) return true;
    const int bytes = blk_cols * 4;
    return ((bytes + 1023) / 1024) * 1024 <= ((bytes + 255) / 256) * 256;
}

// image bytes of `rows` rows of an operand: contiguous run (unsplit) or one run per row (split), rounded to whole pieces
static inline void operand_image(int rows, int ld, int blk_cols, bool split, bool x4, int* ls, int* nseg, int* pps, int* lstride, int* img) {
    const int unit = x4 ? 1024 : 256;
    if (!split) {
        *ls = ld; *nseg = 1; *pps = ceil_div(rows * ld * 4, unit); *lstride = 0; *img = *pps * unit;
    } else {
        *pps = ceil_div(blk_cols * 4, unit); *lstride = *pps * unit; *ls = *lstride / 4; *nseg = rows; *img = rows * *lstride;
    }
}

// Pick block split, wave arrangement, tiles per wave and rows per chunk for (out x in) from the shape alone (deterministic:
// the same shape always runs the same summation order).  The model: matrix time of the padded tiles, HBM time of the operand
// reads (an operand split nb ways on the OTHER side is read nb times), a fixed cost per wave and chunk, one barrier per
// chunk, the reduction's traffic; checked against exhaustive sweeps on the machine (tests/tools/bench_dw_general.py --sweep,
// profiles/r05_dw_general.json: within 0 - 10 % of the best geometry on every shape tried).
static bool plan_dw_g(int out, int lda, int in, int ldb, bool a_x4, bool b_x4, int cus, DwGPlan* best) {
    const int OA = ceil_div(out, 16), OB = ceil_div(in, 16);
    static const int arr[][3] = {{4, 2, 1}, {2, 4, 1}, {8, 1, 1}, {1, 8, 1}, {2, 2, 2}, {4, 1, 2}, {1, 4, 2}, {2, 1, 4}, {1, 2, 4},
                                 {1, 1, 8}};
    bool found = false;
    int force[8] = {0};
    bool forced = false;
    if (const char* f = getenv("NM_DW_FORCE"))       // tuning hook: "nba,nbb,wa,wb,wk,rows"
        forced = sscanf(f, "%d,%d,%d,%d,%d,%d", &force[0], &force[1], &force[2], &force[3], &force[4], &force[5]) == 6;
    for (int nba = 1; nba <= 8; ++nba)
        for (int nbb = 1; nbb <= 8; ++nbb) {
            if (nba * nbb > cus) continue;
            const int OAb = ceil_div(OA, nba), OBb = ceil_div(OB, nbb);
            if ((nba > 1 && ceil_div(OA, OAb) != nba) || (nbb > 1 && ceil_div(OB, OBb) != nbb)) continue;   // empty blocks
            for (const auto& w : arr) {
                const int wa = w[0], wb = w[1], wk = w[2];
                const int ta = ceil_div(OAb, wa), tb = ceil_div(OBb, wb);
                if (ta > DWG_MAX_TA || tb > DWG_MAX_TB || ta * tb > DWG_MAX_TILES) continue;
                if ((ta - 1) * wa >= OAb && wa > 1) continue;      // a narrower arrangement covers the same tiles
                if ((tb - 1) * wb >= OBb && wb > 1) continue;
                for (int rows = 4 * wk < 16 ? 16 : 4 * wk; rows <= 256; rows *= 2) {
                    if (forced && !(nba == force[0] && nbb == force[1] && wa == force[2] && wb == force[3] && wk == force[4] && rows == force[5])) continue;
                    int lsa, lsb, nseg, pps, lstride, imga, imgb;
                    operand_image(rows, lda, wa * ta * 16, nba > 1, pick_x4(a_x4, nba > 1, wa * ta * 16), &lsa, &nseg, &pps, &lstride, &imga);
                    const int pieces_a = ceil_div(nseg * pps, 8);
                    operand_image(rows, ldb, wb * tb * 16, nbb > 1, pick_x4(b_x4, nbb > 1, wb * tb * 16), &lsb, &nseg, &pps, &lstride, &imgb);
                    const int pieces_b = ceil_div(nseg * pps, 8);
                    if (4 * (imga + imgb) + DWG_SLACK > DWG_LDS_BYTES) break;
                    if (2 * (pieces_a + pieces_b) > 60) break;     // vmcnt is a 6-bit counter: two chunks in flight
                    // CU cycles per chunk.  A SIMD runs two of the 8 waves; a wave's chunk is kgw k-groups of ta * tb MFMAs
                    // (32 cycles each) + its operand reads + a fixed part (DMA issue, loop control, the counted wait); the chunk
                    // cannot beat its operand bytes at the rate one CU draws from HBM; one barrier.  k-splitting (wk) multiplies
                    // the partials the order-fixed reduction has to read: a per-call cost, weighed at a typical batch of rows.
                    const double kgw = rows / (4.0 * wk);
                    const double wave = 32.0 * ta * tb * kgw + 6.0 * (ta + tb) * kgw + 400.0 + 40.0 * (pieces_a + pieces_b);
                    const double hbm = 4.0 * rows * ((nba > 1 ? wa * ta * 16 : lda) + (nbb > 1 ? wb * tb * 16 : ldb)) / 7.5;
                    const double chunk = (2.0 * wave > hbm ? 2.0 * wave : hbm) + 300.0;
                    const double partial_bytes = (double)(cus / (nba * nbb)) * wk * (nba * wa * ta * 16.0) * (nbb * wb * tb * 16.0) * 4.0;
                    const double reduce = 2.0 * partial_bytes / 2270.0 * cus / 262144.0;       // device cycles -> CU cycles per row
                    const double cost = chunk * nba * nbb / rows + reduce;
                    if (!found || cost < best->cost * 0.999) {
                        *best = DwGPlan{nba, nbb, wa, wb, wk, ta, tb, rows, 0, lsa, lsb, cost};
                        found = true;
                    }
                }
            }
        }
    return found;
}

static bool dw_aligned(const void* p, int ld) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0 && ld % 4 == 0; }

// ---- the same kernel as a plain GEMM with a SHORT contraction and a WIDE output (layer-wise network evaluation,
// nerf_layerwise.hip): C (out x in) = A^T B = sum_{r < rows} A[r][o] * B[r][i] with A (rows x out, ld lda), B (rows x in, ld ldb)
// row-major -- `in` is a batch of samples (up to 2^19), `rows` a layer's input width.  One sample part (grid x = 1: every
// workgroup walks the whole contraction), the output cut into blocks of up to 256 x 256 (grid y / z); the "partial" IS the
// result, (out_pad x in_pad) row-major; the caller's epilogue adds the bias and applies the activation.
DwgGemmGeometry dwg_gemm_geometry(int out, int64_t in) {
    DwgGemmGeometry g;
    const int tiles = ceil_div(out, 16);
    // the block is always 256 samples wide (one 1 KiB DMA piece per row of B): wb * tb = 16 tiles
    if (tiles <= 4) { g.wa = 1; g.wb = 8; g.ta = tiles; g.tb = 2; }  // heads, narrow layers: all 8 waves side by side over the samples
    else if (tiles <= 8) { g.wa = 2; g.wb = 4; g.ta = ceil_div(tiles, 2); g.tb = 4; }
    else if (tiles <= 16) { g.wa = 4; g.wb = 2; g.ta = ceil_div(tiles, 4); g.tb = 8; }
    else { g.wa = 4; g.wb = 2; g.ta = 4; g.tb = 8; }
    g.nba = ceil_div(tiles, g.wa * g.ta);
    const int64_t cols = (int64_t)g.wb * g.tb * 16;
    g.nbb = (int)((in + cols - 1) / cols);
    g.out_pad = g.nba * g.wa * g.ta * 16;
    g.in_pad = (int64_t)g.nbb * cols;
    return g;
}

int dwg_gemm(const float* A, int out, int lda, const float* B, int64_t in, int64_t ldb, int rows, float* partial, hipStream_t stream,
             const DwgEpilogue* epi) {
    NM_REQUIRE(A && B && (partial || epi) && out >= 1 && in >= 1 && rows >= 1 && lda >= out && ldb >= in, "gemm: bad argument");
    const DwgGemmGeometry g = dwg_gemm_geometry(out, in);
    NM_REQUIRE(g.nbb <= 65535 && g.in_pad < (1ll << 31) && ldb < (1ll << 31), "gemm: batch too large");
    NM_REQUIRE(((int64_t)rows + 32) * (ldb > lda ? ldb : lda) * 4 < 0xf0000000ll, "gemm: operand exceeds 32-bit offsets (use a smaller batch)");
    constexpr int ROWS = 16;
    DwGArgs a;
    auto fill = [&](DwOperand& op, const float* base, int64_t ld, int blk_cols, bool split) {
        op.base = base; op.bytes = (int64_t)rows * ld * 4; op.ld = (int)ld; op.x4 = dw_aligned(base, (int)ld);
        operand_image(ROWS, (int)ld, blk_cols, split, op.x4, &op.ls, &op.nseg, &op.pps, &op.lstride, &op.img);
        op.gstride = (int)(ld * 4); op.blk_cols = split ? blk_cols : 0;
    };
    fill(a.A, A, lda, g.wa * g.ta * 16, g.nba > 1 || lda > 256);
    fill(a.B, B, ldb, g.wb * g.tb * 16, true);
    a.n = rows; a.rows = ROWS; a.wa = g.wa; a.wb = g.wb; a.wk = 1;
    a.out_pad = g.out_pad; a.in_pad = (int)g.in_pad;
    a.partial = partial;
    a.partial_bias = nullptr;
    a.epi_out = epi ? epi->out : nullptr; a.epi_bias = epi ? epi->bias : nullptr; a.epi_mask = epi ? epi->mask : nullptr;
    a.epi_ld = epi ? epi->ld : 0; a.epi_rows = out; a.epi_act = epi ? epi->act : 0;
    NM_REQUIRE(!epi || (epi->out && epi->ld >= g.in_pad && epi->ld % 4 == 0), "gemm: the epilogue's planes must span the padded batch");
    const int lds_bytes = 4 * (a.A.img + a.B.img) + DWG_SLACK;
    NM_REQUIRE(lds_bytes <= DWG_LDS_BYTES, "gemm: LDS budget exceeded");
    DwGKernel kernel = g_dwg_kernels[g.ta - 1][g.tb - 1];
    if (DwGKernel vec = dwg_vec_kernel(g.ta, g.tb))
        if (dwg_vec_ok(a)) kernel = vec;
    if (int rc = ensure_dynamic_lds((const void*)kernel, lds_bytes)) return rc;
    DwGBatch batch;
    batch.common = a; batch.jobs = 1; batch.per_job = 1;
    batch.job[0] = DwGJob{A, B, partial, nullptr};
    hipLaunchKernelGGL(kernel, dim3(1, g.nba, g.nbb), dim3(512), lds_bytes, stream, batch);
    NM_HIP_CHECK(hipGetLastError());
    return 0;
}

// the kernels tuned for the shipped configs' shapes (nerf_dw.hip); -1 = "not mine"
int64_t weight_grad_tuned_workspace_bytes(int32_t out_features, int32_t act_stride, int32_t num_cus);
int weight_grad_tuned(int device_cus, int jobs, const nm_weight_grad_job* job, int32_t out_features, int32_t act_stride,
                      int32_t in_features, int64_t n, void* d_workspace, hipStream_t stream);
int head_grad_tuned(const float* d_dlast, const float* d_act, int32_t in_features, int64_t n, void* d_workspace, float* d_dw,
                    float* d_dbias, hipStream_t stream);
static bool general_only() { const char* e = getenv("NM_DW_GENERAL"); return e && *e && *e != '0'; }   // A/B hook of the tools

}  // namespace nm

using namespace nm;

extern "C" int64_t nm_weight_grad_workspace_bytes_ex(int32_t out_features, int32_t delta_stride, int32_t in_features,
                                                     int32_t act_stride, int32_t num_cus) {
    if (out_features <= 0 || in_features <= 0 || num_cus <= 0 || delta_stride < out_features || act_stride < in_features) return 0;
    // the plan may depend on the operands' alignment (DMA element size -> LDS image size -> rows per chunk): take the larger
    int64_t need = delta_stride == out_features ? weight_grad_tuned_workspace_bytes(out_features, act_stride, num_cus) : 0;
    for (int mode = 0; mode < 4; ++mode) {
        DwGPlan p;
        if (!plan_dw_g(out_features, delta_stride, in_features, act_stride, mode & 1, mode & 2, num_cus, &p)) continue;
        // what nm_weight_grad_batch launches at most per sub-batch: jobs * per_job <= max(1, cus / (nba * nbb)) sample parts
        const int64_t blocks = (int64_t)p.nba * p.nbb;
        const int64_t parts = (num_cus / blocks > 1 ? num_cus / blocks : 1) * p.wk;
        const int64_t out_pad = (int64_t)p.nba * p.wa * p.ta * 16, in_pad = (int64_t)p.nbb * p.wb * p.tb * 16;
        const int64_t bytes = parts * (out_pad * in_pad + out_pad) * 4;
        need = bytes > need ? bytes : need;
    }
    return need;
}

// The general form of nm_weight_grad: d_delta (n, >= out_features) with row stride delta_stride, d_act (n, >= in_features) with
// row stride act_stride (floats), any n >= 1, any widths the planner can tile (up to 8 blocks of 8 x 6 / 8 x 8 tiles per side).
// `jobs` products of ONE shape, stride pair and row count in one launch + one reduction (the same-shape layers of a
// network): a job gets 1 / jobs of the CUs and jobs times the samples per workgroup -- the same matrix work, jobs times fewer
// partials to write and reduce.  The tuned kernel when it serves the shape, the general one otherwise.
// The workspace of ONE product (nm_weight_grad_workspace_bytes_ex) holds max(1, cus / (nba * nbb)) sample parts: when the output
// is cut into so many blocks that `jobs` products no longer fit beside each other (8 x 8 blocks of a 2048-wide layer: 4 parts),
// the batch goes out in sub-batches of that many products, one after the other on the stream, each re-using the workspace.
extern "C" int nm_weight_grad_batch(int device_cus, int32_t jobs, const nm_weight_grad_job* job, int32_t out_features,
                                    int32_t delta_stride, int32_t in_features, int32_t act_stride, int64_t n, void* d_workspace,
                                    void* stream_) {
    NM_REQUIRE(job && jobs >= 1 && jobs <= DWG_MAX_JOBS && d_workspace && n > 0, "bad argument");
    NM_REQUIRE(out_features >= 1 && in_features >= 1 && delta_stride >= out_features && act_stride >= in_features,
               "weight_grad: a row stride is smaller than its feature count");
    bool a_x4 = true, b_x4 = true;
    for (int j = 0; j < jobs; ++j) {
        NM_REQUIRE(job[j].d_delta && job[j].d_act && job[j].d_dw, "weight_grad: a job lacks an operand or its output");
        NM_REQUIRE((reinterpret_cast<uintptr_t>(job[j].d_delta) & 3) == 0 && (reinterpret_cast<uintptr_t>(job[j].d_act) & 3) == 0, "weight_grad: unaligned operand");
        a_x4 = a_x4 && dw_aligned(job[j].d_delta, delta_stride);
        b_x4 = b_x4 && dw_aligned(job[j].d_act, act_stride);
    }
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (delta_stride == out_features && !general_only()) {
        const int rc = weight_grad_tuned(device_cus, jobs, job, out_features, act_stride, in_features, n, d_workspace, stream);
        if (rc >= 0) return rc;
    }
    const int cus = device_cus > 0 ? device_cus : 256;
    DwGPlan p;
    NM_REQUIRE(plan_dw_g(out_features, delta_stride, in_features, act_stride, a_x4, b_x4, cus, &p),
               "weight_grad: no tiling for this (out, in) shape (rows wider than the LDS ring holds)");
    const int fit = cus / (p.nba * p.nbb) > 1 ? cus / (p.nba * p.nbb) : 1;      // sample parts the workspace holds
    if (jobs > fit) {
        for (int j0 = 0; j0 < jobs; j0 += fit)
            if (int rc = nm_weight_grad_batch(device_cus, jobs - j0 < fit ? jobs - j0 : fit, job + j0, out_features, delta_stride,
                                              in_features, act_stride, n, d_workspace, stream_))
                return rc;
        return 0;
    }
    const int64_t chunks = (n + p.rows - 1) / p.rows;
    int64_t per_job = fit / jobs;
    per_job = per_job > chunks ? chunks : per_job;
    NM_REQUIRE((chunks / per_job + 2) * p.rows * (int64_t)(delta_stride > act_stride ? delta_stride : act_stride) * 4 < 0xf0000000ll,
               "weight_grad: a workgroup's sample range exceeds 32-bit offsets");
    DwGBatch batch;
    DwGArgs& a = batch.common;
    auto fill = [&](DwOperand& op, int ld, int blk_cols, bool split, bool x4) {
        op.base = nullptr; op.bytes = n * (int64_t)ld * 4; op.ld = ld; op.x4 = x4;
        operand_image(p.rows, ld, blk_cols, split, x4, &op.ls, &op.nseg, &op.pps, &op.lstride, &op.img);
        op.gstride = ld * 4; op.blk_cols = split ? blk_cols : 0;
    };
    fill(a.A, delta_stride, p.wa * p.ta * 16, p.nba > 1, pick_x4(a_x4, p.nba > 1, p.wa * p.ta * 16));
    fill(a.B, act_stride, p.wb * p.tb * 16, p.nbb > 1, pick_x4(b_x4, p.nbb > 1, p.wb * p.tb * 16));
    a.n = n; a.rows = p.rows; a.wa = p.wa; a.wb = p.wb; a.wk = p.wk;
    a.epi_out = nullptr; a.epi_bias = nullptr; a.epi_mask = nullptr; a.epi_ld = 0; a.epi_rows = 0; a.epi_act = 0;
    a.out_pad = p.nba * p.wa * p.ta * 16; a.in_pad = p.nbb * p.wb * p.tb * 16;
    a.partial = nullptr; a.partial_bias = nullptr;
    const int parts = (int)per_job * p.wk;
    const int64_t job_floats = (int64_t)parts * ((int64_t)a.out_pad * a.in_pad + a.out_pad);
    batch.jobs = jobs; batch.per_job = (int)per_job;
    DwGReduceBatch rb;
    for (int j = 0; j < jobs; ++j) {
        float* partial = static_cast<float*>(d_workspace) + j * job_floats;
        float* partial_bias = partial + (int64_t)parts * a.out_pad * a.in_pad;
        batch.job[j] = DwGJob{job[j].d_delta, job[j].d_act, partial, partial_bias};
        rb.job[j] = DwGReduceJob{partial, partial_bias, job[j].d_dw, job[j].d_dbias, job[j].dw_ld, job[j].dw_col0};
    }
    const int lds_bytes = 4 * (a.A.img + a.B.img) + DWG_SLACK;
    NM_REQUIRE(lds_bytes <= DWG_LDS_BYTES, "weight_grad: LDS budget exceeded");
    DwGKernel kernel = g_dwg_kernels[p.ta - 1][p.tb - 1];
    if (DwGKernel vec = dwg_vec_kernel(p.ta, p.tb))
        if (dwg_vec_ok(a)) kernel = vec;
    if (int rc = ensure_dynamic_lds((const void*)kernel, lds_bytes)) return rc;
    hipLaunchKernelGGL(kernel, dim3((unsigned)(per_job * jobs), p.nba, p.nbb), dim3(512), lds_bytes, stream, batch);
    const int64_t elems = (int64_t)out_features * in_features;
    hipLaunchKernelGGL(dw_reduce_g_kernel, dim3((unsigned)((elems + out_features + 255) / 256), jobs), dim3(256), 0, stream, rb, parts,
                       (int64_t)a.out_pad * a.in_pad, a.out_pad, out_features, a.in_pad, in_features);
    NM_HIP_CHECK(hipGetLastError());
    return 0;
}

// one product = a batch of one
extern "C" int nm_weight_grad_ex(int device_cus, const float* d_delta, int32_t out_features, int32_t delta_stride,
                                 const float* d_act, int32_t in_features, int32_t act_stride, int64_t n, void* d_workspace,
                                 float* d_dw, int32_t dw_ld, int32_t dw_col0, float* d_dbias, void* stream_) {
    NM_REQUIRE(d_delta && d_act && d_workspace && d_dw && n > 0, "bad argument");
    const nm_weight_grad_job job = {d_delta, d_act, d_dw, dw_ld, dw_col0, d_dbias};
    return nm_weight_grad_batch(device_cus, 1, &job, out_features, delta_stride, in_features, act_stride, n, d_workspace, stream_);
}

// what the planner chose for a shape (tools / tests): [nba, nbb, wa, wb, wk, ta, tb, rows]
extern "C" int nm_weight_grad_plan(int32_t out_features, int32_t delta_stride, int32_t in_features, int32_t act_stride,
                                   int32_t aligned16, int32_t num_cus, int32_t* plan8) {
    NM_REQUIRE(plan8, "bad argument");
    DwGPlan p;
    NM_REQUIRE(plan_dw_g(out_features, delta_stride, in_features, act_stride, aligned16 && delta_stride % 4 == 0,
                         aligned16 && act_stride % 4 == 0, num_cus > 0 ? num_cus : 256, &p), "weight_grad: no tiling for this shape");
    const int v[8] = {p.nba, p.nbb, p.wa, p.wb, p.wk, p.ta, p.tb, p.rows};
    for (int k = 0; k < 8; ++k) plan8[k] = v[k];
    return 0;
}

extern "C" int64_t nm_head_grad_workspace_bytes_ex(int32_t in_features) {
    return in_features > 0 ? (int64_t)HEAD_G_MAX_PARTS * 4 * ((int64_t)in_features + 1) * 4 : 0;
}

// d_dlast (n, 4) contiguous, d_act (n, >= in_features) with row stride act_stride: any width.
extern "C" int nm_head_grad_ex(const float* d_dlast, const float* d_act, int32_t in_features, int32_t act_stride, int64_t n,
                               void* d_workspace, float* d_dw, float* d_dbias, void* stream_) {
    NM_REQUIRE(d_dlast && d_act && d_workspace && d_dw && n > 0 && in_features >= 1 && act_stride >= in_features, "bad argument");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (act_stride == in_features && !general_only()) {
        const int rc = head_grad_tuned(d_dlast, d_act, in_features, n, d_workspace, d_dw, d_dbias, stream);
        if (rc >= 0) return rc;
    }
    const bool vec4 = in_features % 4 == 0 && dw_aligned(d_act, act_stride);
    const int vec = vec4 ? 4 : 1;
    const int cols = (in_features + vec - 1) / vec;              // column slots
    const int tpr = cols < 256 ? cols : 256;                     // threads per row
    const int groups = (cols + tpr - 1) / tpr;                   // column groups (grid.y): widths beyond 256 slots
    const int rpp = 256 / tpr;
    int64_t rows = (n + HEAD_G_MAX_PARTS - 1) / HEAD_G_MAX_PARTS;
    rows = (rows + 8 * rpp - 1) / (8 * rpp) * (8 * rpp);
    const int parts = (int)((n + rows - 1) / rows);
    float* partial = static_cast<float*>(d_workspace);
    float* partial_bias = partial + (int64_t)HEAD_G_MAX_PARTS * 4 * in_features;
    const int lds_bytes = (rpp * 4 * tpr * vec + rpp * 4) * 4;
    if (vec4)
        hipLaunchKernelGGL(head_grad_g_kernel<4>, dim3(parts, groups), dim3(256), lds_bytes, stream, d_dlast, d_act, act_stride,
                           in_features, tpr, n, (int)rows, partial, partial_bias);
    else
        hipLaunchKernelGGL(head_grad_g_kernel<1>, dim3(parts, groups), dim3(256), lds_bytes, stream, d_dlast, d_act, act_stride,
                           in_features, tpr, n, (int)rows, partial, partial_bias);
    const int elems = 4 * in_features;
    hipLaunchKernelGGL(head_reduce_g_kernel, dim3((elems + 4 + 63) / 64), dim3(256), 0, stream, partial, partial_bias, parts,
                       elems, d_dw, d_dbias);
    NM_HIP_CHECK(hipGetLastError());
    return 0;
}
