// Back-propagation of the 64-wide networks (BASELINE config 1's 4x64, 8x64) in ONE kernel for gfx950: the delta chain of
// nerf_train.hip's mlp_backward_kernel AND every weight / bias gradient of
//   /root/reference/src/nerf/models.py:60-80 (what autograd's addmm backward computes for layer1, layers_xyz[*], fc_feat,
//   layers_dir[0], fc_alpha, fc_rgb) under /root/reference/src/models/model_nerf.py:88-151 (training_step + loss.backward()),
// so that no delta row is ever written to HBM and the tape is read exactly once.
//
// Why only 64 wide.  dW = delta^T @ act contracts over the samples: its accumulators must stay resident while a workgroup walks
// through its samples, for ALL layers at once because the chain produces one layer's delta after the other per sample tile.  A
// 64 x 64 product is 16 KB: the 7 - 11 products of a 4 - 8 layer network are 56 - 88 accumulator registers per wave of an
// 8-wave workgroup.  At 128 wide the same set is 630 KB per workgroup -- more than a CU's register file --,
// which is why the wider networks keep the separate delta and weight-gradient kernels.
//
// Dataflow of one workgroup iteration (128 samples = 8 waves x one 16-sample MFMA tile):
//   * the delta chain as in mlp_backward_kernel: a wave's tile stays in registers (D layout of stage k == B layout of stage
//     k + 1), the transposed weights stream L2 -> LDS through the 2-slot ring of mlp_device.h, ReLU' from the taped bit masks;
//   * each delta, once masked, is ALSO written to an LDS tile [sample][feature] (planar and swizzled: fb_write_delta -- the
//     weight-gradient products' 4-byte operand reads are conflict-free);
//   * the activation rows that delta contracts with (tape_h[i] / tape_feat / the encoding rows: 128 consecutive 256-byte rows =
//     one contiguous 32 KB run) are DMA'd HBM -> LDS by scalar-addressed buffer_load ... lds one delta ahead into a 3-slot ring;
//   * every wave owns 2 of the 16 output tiles of each 64 x 64 product (1 of the 8 tiles of the 32 x 64 view-layer products)
//     and contracts them over all 128 samples: v_mfma_f32_16x16x4_f32 with the sample index as the instruction's contraction
//     index, A = one float of the delta tile, B = two floats of the activation row per lane and k-group (dw_kernel's feature
//     permutation: tile q row i stands for feature 4 i + q);  bias gradients are the column sums of the A operands;
//   * the two 4-row heads ride along: d_last is a small LDS tile of its own, fc_alpha = d_last^T h[L-1] on the row block of
//     fc_feat's product, fc_rgb = d_last^T v on the columns 32..63 of the direction-encoding rows, where the taping forward puts
//     the view layer's activation rows for these networks (nm_mlp_tape.v_stride).
// Phases (one barrier each), for delta k:  A_k = [write delta_k to LDS | first weight chunk of stage k],  B_k = [second chunk |
// the dW products of delta_k | DMA of the rows delta_k+1 needs].  Waits are COUNTED (s_waitcnt vmcnt(n) in front of a bare
// s_barrier): the activation rows come from HBM and get a whole stage (~3 us) to land.
// At the end a workgroup writes ONE partial of every product; fb_reduce_kernel adds the partials in index order (deterministic).
// Where the time goes, and what was measured and dropped (deeper DMA prefetch, DMA from inline assembly, ReLU' off the LDS rows,
// a row-major delta tile, the fully unrolled operand loop): DESIGN.md 3.5; tests/tools/probes/fb_probe.hip compiles parts of this
// kernel out (the ABL template parameter) and times the rest.
//
// Roofline: MFMA (delta chain + weight gradients: 2 x the forward's FLOP); HBM traffic = the tape once (2 KB / sample).
#include <cstdlib>

#include "nm_internal.h"
#include "mlp_device.h"

namespace nm {

// ReLU' from a lane's taped bit mask: acc where bit (4 nt + r) is set, +0.0f elsewhere -- v_bfe_i32 (the bit, sign-extended: 0 or ~0)
// and v_and_b32 on the value's bits: two VALU instructions per value (and / compare / select is three; VALU issue time adds to
// matrix time in these kernels)
__device__ __forceinline__ void fb_apply_mask(float (&in)[16], const f32x4 (&acc)[4], const uint64_t m) {
    const int lo = (int)(unsigned)m;                 // 16 tile registers: bits 0..15
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            in[4 * nt + r] = __int_as_float(__float_as_int(acc[nt][r]) & __builtin_amdgcn_sbfe(lo, 4 * nt + r, 1));
}

constexpr int FB_ROWS = 128;                       // samples per workgroup iteration
constexpr int FB_ROWB = 256;                       // bytes per 64-float row
constexpr int FB_CHUNK = 8192;                     // one weight chunk: 8 k-steps x 4 tiles x 256 B
constexpr int FB_OFF_DBUF = 2 * FB_CHUNK;          // the delta tile behind the 2-slot weight ring
constexpr int FB_SLOT = FB_ROWS * FB_ROWB;         // 32 KB: a block of activation rows
constexpr int FB_NSLOT = 3;
constexpr int FB_OFF_SLOT = FB_OFF_DBUF + FB_ROWS * FB_ROWB;
constexpr int FB_OFF_HEADS = FB_OFF_SLOT + FB_NSLOT * FB_SLOT;
constexpr int FB_OFF_DL = FB_OFF_HEADS + 1024;     // the heads' delta tile: 128 rows of 16 floats (rgb x3, sigma, twelve zeros)
constexpr int FB_LDS = FB_OFF_DL + FB_ROWS * 64;   // 156 672 B: one workgroup per CU
constexpr int FB_MAXL = 8;
// one workgroup's partial, in floats: [dir x feat 32x64][dir x enc_d 32x64][feat][xyz 0..6][skip][layer1] then the bias sums
constexpr int FB_P_DIRF = 0, FB_P_DIRE = 2048, FB_P_FEAT = 4096, FB_P_XYZ = 8192, FB_P_SKIP = FB_P_XYZ + (FB_MAXL - 1) * 4096,
              FB_P_L1 = FB_P_SKIP + 4096, FB_P_BIAS = FB_P_L1 + 4096;
constexpr int FB_B_DIR = 0, FB_B_FEAT = 64, FB_B_XYZ = 128, FB_B_L1 = FB_B_XYZ + (FB_MAXL - 1) * 64;
// the two 4-row heads: [column tile][4 rows][16 columns] -- fc_rgb over the columns 32..63 of the direction rows (2 tiles), fc_alpha
// over the 64 columns of h[L-1] (4 tiles) --, then the column sums of d_last [4]; the waves' k-parts are added up in LDS first
constexpr int FB_P_HEAD = FB_P_BIAS + FB_B_L1 + 64, FB_H_RGB = 0, FB_H_ALPHA = 128, FB_H_BIAS = 384;
constexpr int FB_PART = FB_P_HEAD + FB_H_BIAS + 16;

struct FusedBwdArgs {
    const float* tape_h;      // (L, n, 64)
    const float* tape_feat;   // (n, 64)
    const float* enc_x;       // (n, 64)
    const float* enc_d;       // (n, 64)
    float* partial;           // (grid, FB_PART)
    int32_t skip_layer;       // i of the one layers_xyz[i] that takes cat(x, xyz), -1: none
    const float* w1t;         // (dx + 1, 64): layer1.weight^T, then layer1.bias (nm_mlp_export_layer1_transposed; the epilogue's operand)
    int32_t dx;               // width of the position encoding
};

// 128 rows of 256 B -> one LDS slot: 32 pieces of 1 KiB, 4 per wave (every wave issues the same count: the waits count them)
__device__ __forceinline__ void fb_dma_rows(const float* rows, char* dst, int wave) {
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(rows), (short)16, 0x7fffffff, 1 << 23);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int piece = wave + 8 * j;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(dst + piece * 1024), 16, 0, piece * 1024, 0, 0);
    }
}

// The LDS delta tile, [sample][64 words], is PLANAR: word position of feature f of sample s =
//     (plane << 4 | index) ^ swz(s),   swz(s) = (s & 1) << 4 | ((s >> 1) & 7) << 1,
// with plane = f & 3, index = f >> 2 for a 64-wide delta (plane = f & 1, index = f >> 1 for the 32-wide view delta): the 16 rows
// of an A tile (features 4 i + q, resp. 2 i + q) are 16 consecutive words, and the samples 4 ks + k a ds_read_b32's lane groups
// fetch together (k = 0, 1 | 2, 3) land in different halves of the 32 banks -- the reads are conflict-free (the row-major tile
// of the first version cost 4 LDS cycles per lane group: bank = word mod 32 for 4-byte accesses, MI355X_MICROARCH.md, LDS).
// A lane of the chain holds features 16 nt + 4 g + r of sample `col` of its wave's tile (D layout): 16 (8) ds_write_b32, whose
// 32-lane groups (g & 1, col) cover all 32 banks.
__device__ __forceinline__ unsigned fb_swz(int s) { return ((s & 1) << 4) | (((s >> 1) & 7) << 1); }

template <int NT>
__device__ __forceinline__ void fb_write_delta(char* dbuf, const float (&v)[4 * NT], int wave, int g, int col) {
    float* row = reinterpret_cast<float*>(dbuf + (wave * 16 + col) * FB_ROWB);
    unsigned sw = fb_swz(col);                                        // s & 15 == col
    asm volatile("" : "+v"(sw));      // the 16 word offsets below are recomputed per call (one v_xor each), not kept in 16 registers
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const unsigned pos = NT == 4 ? (unsigned)((r << 4) | (4 * nt + g)) : (unsigned)(((r & 1) << 4) | (8 * nt + 2 * g + (r >> 1)));
            row[pos ^ sw] = v[4 * nt + r];
        }
}

// KS k-steps of a 4-tile stage out of one ring slot (operands of the next k-step in flight, as gemm_stage's PIPE); no barrier
template <int KS, int B0, int NB>
__device__ __forceinline__ void fb_chunk(f32x4 (&acc)[4], const float (&b)[NB], const char* buf) {
    f32x4 a_next = *reinterpret_cast<const f32x4*>(buf);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const f32x4 a = a_next;
        if (ks + 1 < KS) a_next = *reinterpret_cast<const f32x4*>(buf + (ks + 1) * 1024);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q], b[B0 + ks], acc[q], 0, 0, 0);
    }
}

// One weight-gradient product of this wave over the 128 samples of the delta tile: TB = 2 tiles (64 x 64 products) or 1 (the
// 32 x 64 view-layer products).  a_off: this lane's A address for k-groups = 0 (mod 4) -- the swizzle of k-group ks adds an XOR of
// word bits 2..3 with ks & 3 --, b_ptr: this lane's B address in the slot.  Operands are fetched 4 k-groups ahead of their MFMAs.
template <int TB, int ABL = 0, int PF = 4>
__device__ __forceinline__ void fb_dw_step(f32x4 (&acc)[TB], float& bsum, const char* lds, const unsigned a_off, const char* b_ptr) {
    // Batches of 4 k-groups, two per trip of a REAL loop (fully unrolled, the compiler issued all 32 operand reads of a product
    // up front: 60 registers more than the two batches in flight here, and the kernel spilled).
    constexpr int NBATCH = (FB_ROWS / 4) / PF;                 // PF: 4 (2 for the 8-layer instantiation: 12 registers it does not have)
    static_assert(NBATCH % 2 == 0 && (2 * PF) % 4 == 0, "two batches per trip, whole groups of 4 k-groups");
    float a0[PF], a1[PF];
    f32x2 b0[PF], b1[PF];
    const char* ap[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) ap[j] = lds + (a_off ^ (unsigned)(j << 4)) + j * 1024;      // k-group = j (mod 4): swizzle term j << 2 words
    // batch `ph` (0 / 1) of the trip at byte offset trip_off (a multiple of 4 k-groups): k-group q = ph * PF + j of the trip
    auto fetch = [&](float (&a)[PF], f32x2 (&b)[PF], const int ph, const int trip_off) {
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            const int q = ph * PF + j, jj = q & 3;
            if constexpr (ABL & 32) a[j] = 1.0f + j;
            else a[j] = *reinterpret_cast<const float*>(ap[jj] + trip_off + (q - jj) * 1024);
            if constexpr (ABL & 16) b[j] = f32x2{a[j], 2.0f};
            else if constexpr (TB == 2) b[j] = *reinterpret_cast<const f32x2*>(b_ptr + trip_off + q * 1024);
            else b[j][0] = *reinterpret_cast<const float*>(b_ptr + trip_off + q * 1024);
        }
    };
    auto mfmas = [&](const float (&a)[PF], const f32x2 (&b)[PF]) {
#pragma unroll
        for (int j = 0; j < PF; ++j) {
#pragma unroll
            for (int t = 0; t < TB; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j][t], acc[t], 0, 0, 0);
            bsum += a[j];      // column sums of delta = the bias gradient (used from the waves that own column tile 0)
        }
    };
    fetch(a0, b0, 0, 0);
#pragma unroll 1
    for (int off = 0; off < NBATCH * PF * 1024; off += 2 * PF * 1024) {
        fetch(a1, b1, 1, off);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(a0, b0);
        fetch(a0, b0, 0, off + 2 * PF * 1024);   // (the last trip reads the rows behind the block: inside the LDS allocation, never used)
        __builtin_amdgcn_sched_barrier(0);
        mfmas(a1, b1);
    }
}

// A 4-row head product (fc_rgb / fc_alpha, models.py:71,75): A = the heads' delta tile (row s = 16 floats, 4 live), B = 16 natural
// columns of a row block; this wave takes the k-groups kpar, kpar + KS, ...: 32 / KS MFMAs on one accumulator.  a_ptr / b_ptr: this
// lane's operand addresses of k-group kpar.
template <int KS>
__device__ __forceinline__ void fb_head_step(f32x4& acc, float& bsum, const char* a_ptr, const char* b_ptr) {
    constexpr int N = (FB_ROWS / 4) / KS;
#pragma unroll
    for (int q0 = 0; q0 < N; q0 += 4) {
        float a[4], b[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            a[q] = *reinterpret_cast<const float*>(a_ptr + (q0 + q) * (KS * 256));
            b[q] = *reinterpret_cast<const float*>(b_ptr + (q0 + q) * (KS * 1024));
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q], b[q], acc, 0, 0, 0);
            bsum += a[q];
        }
    }
}

template <int ABL = 0>
__device__ __forceinline__ void fb_wait_barrier(int pieces_in_flight, bool phase_a = false) {
    // everything but the newest `pieces_in_flight` DMA pieces (the rows of the NEXT delta) has landed; those stay in flight
    if constexpr (ABL & 128) return;                     // probe: neither waits nor barriers
    if constexpr (ABL & 64) { if (phase_a) return; }     // probe: no barrier between the two phases of a delta
    if constexpr (ABL & 256) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); return; }   // probe: barriers, no DMA waits
    if (pieces_in_flight >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (pieces_in_flight >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

// MAXL: the num_layers an instantiation holds accumulators for (4: 56 registers per wave, 8: 88); the partial layout is MAXL 8's
// ABL: timing ablations of tests/tools/probes/fb_probe.hip (1: no dW products, 2: no row DMA, 4: no chain MFMAs, 8: no delta tile, 16 / 32: no B / A operand reads, 64: no barrier between a delta's phases, 128: no waits or barriers, 256: barriers but no DMA waits); 0 in the library
template <int MAXL, int ABL = 0>
__global__ __launch_bounds__(512, 1) void mlp_backward_dw64_kernel(const MlpBwdArgs args, const FusedBwdArgs fa, const int L) {
    constexpr int H = 64, KH = 16, KD = 8;
    constexpr int DPF = MAXL > 4 ? 2 : 4;                  // operand batches of the dW products (registers: fb_dw_step)
    extern __shared__ __attribute__((aligned(16))) char lds[];
    float* lds_walpha = reinterpret_cast<float*>(lds + FB_OFF_HEADS);   // [4][H/4]
    float* lds_wrgb = lds_walpha + H;                                   // [3][4][H/8]
    for (int i = threadIdx.x; i < H; i += 512) lds_walpha[i] = args.walpha[i];
    for (int i = threadIdx.x; i < 3 * H / 2; i += 512) lds_wrgb[i] = args.wrgb[i];
    for (int i = threadIdx.x; i < FB_ROWS * 16; i += 512) reinterpret_cast<float*>(lds + FB_OFF_DL)[i] = 0.0f;   // columns 4..15 stay zero
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, col = lane & 15;
    char* const dbuf = lds + FB_OFF_DBUF;

    // ---- this wave's output tiles and the operand addresses that go with them
    const int qa = wave >> 1, qb0 = 2 * (wave & 1);        // 64 x 64: tiles (qa, qb0), (qa, qb0 + 1); tile q row i = feature 4 i + q
    const int qa2 = wave >> 2, qb2 = wave & 3;             // 32 x 64: tile (qa2, qb2); A row i = feature 2 i + qa2
    // A of k-group ks: sample 4 ks + g, word (q << 4 | col) ^ swz(4 ks + g); swz = (g & 1) << 4 | (g >> 1) << 1 here, (ks & 3) << 2 in fb_dw_step
    const unsigned a_off64 = FB_OFF_DBUF + g * FB_ROWB + 4 * ((unsigned)((qa << 4) | col) ^ (unsigned)(((g & 1) << 4) | ((g >> 1) << 1)));
    const unsigned a_off32 = FB_OFF_DBUF + g * FB_ROWB + 4 * ((unsigned)((qa2 << 4) | col) ^ (unsigned)(((g & 1) << 4) | ((g >> 1) << 1)));
    const int b_off64 = g * FB_ROWB + col * 16 + qb0 * 4;
    const int b_off32 = g * FB_ROWB + col * 16 + qb2 * 4;
    const bool bias64 = (wave & 1) == 0, bias32 = qb2 == 0;
    // the two waves of a SIMD (waves w and w + 4) take a phase B in opposite orders -- one runs the chain's second weight chunk
    // (matrix-dense, one LDS read per 4 MFMAs) while the other runs the products (an operand pair per 2 MFMAs + the bias sums).
    // Measured: no faster than both in the same order (0.531 ms either way; split by wave & 1 instead: 0.548), but the allocator
    // needs 18 registers less, which frees the 8-layer instantiation of its last spill
    const bool dw_first = (wave & 4) != 0;
    // the heads: fc_rgb = natural column tile 2 + (wave & 1) of the direction rows, k-groups = wave >> 1 (mod 4); fc_alpha = column tile
    // wave & 3 of h[L-1], k-groups = wave >> 2 (mod 2).  A: row 4 ks + g of the heads' delta tile, word col
    const int hr_t = 2 + (wave & 1), hr_k = wave >> 1, ha_t = wave & 3, ha_k = wave >> 2;
    const char* const dl_lane = lds + FB_OFF_DL + g * 64 + col * 4;
    f32x4 acc_rgb = {0.f, 0.f, 0.f, 0.f}, acc_alpha = acc_rgb;
    float bs_head = 0.f;

    f32x4 acc_dirf[1] = {{0.f, 0.f, 0.f, 0.f}}, acc_dire[1] = {{0.f, 0.f, 0.f, 0.f}};
    f32x4 acc_feat[2], acc_skip[2], acc_l1[2], acc_xyz[MAXL - 1][2];
    float bs_dir = 0.f, bs_feat = 0.f, bs_xyz[MAXL - 1];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        acc_feat[t] = acc_skip[t] = acc_l1[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < MAXL - 1; ++i) acc_xyz[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < MAXL - 1; ++i) bs_xyz[i] = 0.f;

    const int64_t wg_iters = args.n / FB_ROWS;             // n % 128 == 0 (the host checks)
    const int64_t mstride = args.tiles * 64;
    const int sk = fa.skip_layer;
    int par = 0;            // ring slot of the weight chunk the next gemm phase consumes
    int bslot = 0;          // activation slot of the next product
    int bissue = 0;         // activation slot the next DMA goes to
    auto slot_ptr = [&](int s) { return lds + FB_OFF_SLOT + s * FB_SLOT; };
    auto next_slot = [](int s) { return s == FB_NSLOT - 1 ? 0 : s + 1; };
    auto issue_rows = [&](const float* rows) { if constexpr (!(ABL & 2)) fb_dma_rows(rows, slot_ptr(bissue), wave); bissue = next_slot(bissue); };

    int64_t it = blockIdx.x;
    f32x4 go = {0.f, 0.f, 0.f, 0.f}, y = go;
    uint64_t mv = 0;
    auto fetch_head = [&](int64_t iter) {
        const int64_t s = (iter * 8 + wave) * 16 + col;
        go = *reinterpret_cast<const f32x4*>(args.grad_out + 4 * s);
        y = *reinterpret_cast<const f32x4*>(args.radiance + 4 * s);
        mv = args.mask_v[(iter * 8 + wave) * 64 + lane];
    };
    if (it < wg_iters) {
        fetch_head(it);
        stream_to_lds<8>(args.wstream, lds, FB_CHUNK, wave, lane);                 // layers_dir.0^T: one chunk
        issue_rows(fa.tape_feat + it * (FB_ROWS * 64));
        issue_rows(fa.enc_d + it * (FB_ROWS * 64));
    }
    fb_wait_barrier<ABL>(8);     // the head weights in LDS and the first weight chunk (older than the 8 row pieces) are everybody's now

    for (; it < wg_iters; it += gridDim.x) {
        const bool has_next = it + gridDim.x < wg_iters;
        const int64_t tile = it * 8 + wave;
        const int64_t sample = tile * 16 + col;
        const int64_t row0 = it * FB_ROWS;                 // first sample of the workgroup's block
        const uint64_t* mrow = args.mask_h + tile * 64 + lane;
        const char* gw = args.wstream;

        // ================= delta 0: at layers_dir[0]'s pre-activation (sigmoid', fc_rgb^T on the VALU; models.py:74-75)
        float drgb[3];
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) drgb[ch] = go[ch] * (y[ch] * (1.0f - y[ch]));
        const float dsigma = go[3];
        if (g == 0) {
            const f32x4 o4 = {drgb[0], drgb[1], drgb[2], dsigma};
            if (args.d_last) *reinterpret_cast<f32x4*>(args.d_last + 4 * sample) = o4;
            *reinterpret_cast<f32x4*>(lds + FB_OFF_DL + (wave * 16 + col) * 64) = o4;      // visible behind phase A0's barrier
        }
        float dv[KD];
#pragma unroll
        for (int s = 0; s < KD; ++s) {
            float a = lds_wrgb[(0 * 4 + g) * KD + s] * drgb[0];
            a = fmaf(lds_wrgb[(1 * 4 + g) * KD + s], drgb[1], a);
            a = fmaf(lds_wrgb[(2 * 4 + g) * KD + s], drgb[2], a);
            dv[s] = ((mv >> s) & 1u) ? a : 0.0f;
        }
        f32x4 acc[4];
        float in[KH];
        // ---- phase A0: delta_v -> LDS; layers_dir.0^T (hidden columns), one chunk
        {
            const uint64_t m = mrow[(int64_t)(L - 1) * mstride];
            if constexpr (!(ABL & 8)) fb_write_delta<2>(dbuf, dv, wave, g, col);
            stream_to_lds<8>(gw + FB_CHUNK, lds + (par ^ 1) * FB_CHUNK, FB_CHUNK, wave, lane);      // fc_feat^T chunk 0
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (!(ABL & 4)) fb_chunk<KD, 0, KD>(acc, dv, lds + par * FB_CHUNK + lane * 16);
            fb_wait_barrier<ABL>(0, true);
            par ^= 1;
            gw += FB_CHUNK;
            fb_apply_mask(in, acc, m);
        }
        // ---- phase B0: grad(layers_dir.0) = delta_v^T @ [feat | view encoding] (models.py:72-73)
        {
            issue_rows(fa.tape_h + ((int64_t)(L - 1) * args.n + row0) * 64);
            if constexpr (!(ABL & 1)) fb_dw_step<1, ABL, DPF>(acc_dirf, bs_dir, lds, a_off32, slot_ptr(bslot) + b_off32);
            bslot = next_slot(bslot);
            float unused = 0.f;
            if constexpr (!(ABL & 1)) fb_dw_step<1, ABL, DPF>(acc_dire, unused, lds, a_off32, slot_ptr(bslot) + b_off32);
            // grad(fc_rgb) = d_last^T @ v: the view layer's activation rows are the columns 32..63 of the direction rows (models.py:75)
            if constexpr (!(ABL & 1)) fb_head_step<4>(acc_rgb, unused, dl_lane + hr_k * 256, slot_ptr(bslot) + hr_k * 1024 + g * FB_ROWB + (16 * hr_t + col) * 4);
            bslot = next_slot(bslot);
            fb_wait_barrier<ABL>(4);
        }
        // ================= delta 1: at fc_feat's pre-activation; fc_feat^T + fc_alpha^T (models.py:70-71)
        {
            const uint64_t m = mrow[(int64_t)(L - 2) * mstride];
            // ---- phase A1
            if constexpr (!(ABL & 8)) fb_write_delta<4>(dbuf, in, wave, g, col);
            stream_to_lds<8>(gw + FB_CHUNK, lds + (par ^ 1) * FB_CHUNK, FB_CHUNK, wave, lane);
            const float* wa = lds_walpha + g * (H / 4);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const f32x4 w4 = *reinterpret_cast<const f32x4*>(wa + 4 * nt);
                acc[nt] = f32x4{w4[0] * dsigma, w4[1] * dsigma, w4[2] * dsigma, w4[3] * dsigma};
            }
            if constexpr (!(ABL & 4)) fb_chunk<8, 0, KH>(acc, in, lds + par * FB_CHUNK + lane * 16);
            fb_wait_barrier<ABL>(0, true);
            par ^= 1;
            // ---- phase B1: the first chunk of the next chain stage -- layers_xyz[L-2]^T, or (L = 2: layers_xyz[0]^T is never applied,
            //      see the last delta) of this workgroup's next iteration | rows of delta 2
            const bool next_chain = L >= 3;
            if (next_chain || has_next) stream_to_lds<8>(next_chain ? gw + 2 * FB_CHUNK : args.wstream, lds + (par ^ 1) * FB_CHUNK, FB_CHUNK, wave, lane);
            int flying = 4;
            if (L == 2) {
                issue_rows(fa.enc_x + row0 * 64);          // the last delta contracts with the encoding rows only (see there)
            } else {
                issue_rows(fa.tape_h + ((int64_t)(L - 2) * args.n + row0) * 64);
                if (sk == L - 2) { issue_rows(fa.enc_x + row0 * 64); flying = 8; }
            }
            if constexpr (!(ABL & 4)) if (!dw_first) fb_chunk<8, 8, KH>(acc, in, lds + par * FB_CHUNK + lane * 16);
            if constexpr (!(ABL & 1)) fb_dw_step<2, ABL, DPF>(acc_feat, bs_feat, lds, a_off64, slot_ptr(bslot) + b_off64);
            // grad(fc_alpha) = d_last^T @ h[L-1] (row 3; models.py:71), and the column sums of d_last = both heads' bias gradients
            if constexpr (!(ABL & 1)) fb_head_step<2>(acc_alpha, bs_head, dl_lane + ha_k * 256, slot_ptr(bslot) + ha_k * 1024 + g * FB_ROWB + (16 * ha_t + col) * 4);
            if constexpr (!(ABL & 4)) if (dw_first) fb_chunk<8, 8, KH>(acc, in, lds + par * FB_CHUNK + lane * 16);
            bslot = next_slot(bslot);
            fb_wait_barrier<ABL>(flying);
            par ^= 1;
            gw += 2 * FB_CHUNK;
            fb_apply_mask(in, acc, m);
        }
        // ================= deltas 2 .. L-1: at layers_xyz[i]'s pre-activation, i = L-2 .. 1; layers_xyz[i]^T (models.py:63-69)
#pragma unroll
        for (int i = MAXL - 2; i >= 1; --i) {
            if (i <= L - 2) {
                const uint64_t m = mrow[(int64_t)(i - 1) * mstride];
                // ---- phase A
                if constexpr (!(ABL & 8)) fb_write_delta<4>(dbuf, in, wave, g, col);
                stream_to_lds<8>(gw + FB_CHUNK, lds + (par ^ 1) * FB_CHUNK, FB_CHUNK, wave, lane);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
                if constexpr (!(ABL & 4)) fb_chunk<8, 0, KH>(acc, in, lds + par * FB_CHUNK + lane * 16);
                fb_wait_barrier<ABL>(0, true);
                par ^= 1;
                // ---- phase B: first chunk of layers_xyz[i-1]^T -- or, behind layers_xyz[1]^T, of the next iteration | rows of the next delta
                if (i > 1 || has_next) stream_to_lds<8>(i > 1 ? gw + 2 * FB_CHUNK : args.wstream, lds + (par ^ 1) * FB_CHUNK, FB_CHUNK, wave, lane);
                int flying = 4;
                if (i == 1) {
                    issue_rows(fa.enc_x + row0 * 64);      // the last delta contracts with the encoding rows only
                } else {
                    issue_rows(fa.tape_h + ((int64_t)(i - 1) * args.n + row0) * 64);
                    if (sk == i - 1) { issue_rows(fa.enc_x + row0 * 64); flying = 8; }
                }
                if constexpr (!(ABL & 4)) if (!dw_first) fb_chunk<8, 8, KH>(acc, in, lds + par * FB_CHUNK + lane * 16);
                if constexpr (!(ABL & 1)) fb_dw_step<2, ABL, DPF>(acc_xyz[i], bs_xyz[i], lds, a_off64, slot_ptr(bslot) + b_off64);
                bslot = next_slot(bslot);
                if (sk == i) {                                             // cat(x, xyz): the encoding columns (models.py:64-65)
                    float unused = 0.f;
                    if constexpr (!(ABL & 1)) fb_dw_step<2, ABL, DPF>(acc_skip, unused, lds, a_off64, slot_ptr(bslot) + b_off64);
                    bslot = next_slot(bslot);
                }
                if constexpr (!(ABL & 4)) if (dw_first) fb_chunk<8, 8, KH>(acc, in, lds + par * FB_CHUNK + lane * 16);
                fb_wait_barrier<ABL>(flying);
                par ^= 1;
                gw += 2 * FB_CHUNK;
                fb_apply_mask(in, acc, m);
            }
        }
        // ================= the last delta: at layers_xyz[0]'s pre-activation.  layer1 has NO activation (models.py:62), so (a) the delta
        // at its output is LINEAR in this one -- W0^T delta per sample -- and (b) the activation rows this delta's own product would
        // contract with are linear in the encoding -- h[0] = W1 enc + b1.  With S = [delta^T @ enc | column sums of delta] (64 x 64):
        //     grad(layer1.weight | bias) = W0^T S,           grad(layers_xyz[0].weight) = S [W1 | b1]^T
        // (W0 = layers_xyz[0]'s hidden columns, W1 = layer1.weight).  The chain stops here and this delta has ONE product, with the
        // encoding rows; both matrix products with S run once per workgroup, in the epilogue: per iteration a chain stage (64 of
        // 352 chain MFMAs at 4 layers), a product over the samples (64 of 472), a 32 KB block of rows and two barriers less.
        if constexpr (!(ABL & 8)) fb_write_delta<4>(dbuf, in, wave, g, col);
        fb_wait_barrier<ABL>(0, true);
        int flying = 0;
        if (has_next) {
            fetch_head(it + gridDim.x);
            issue_rows(fa.tape_feat + (it + gridDim.x) * (FB_ROWS * 64));
            issue_rows(fa.enc_d + (it + gridDim.x) * (FB_ROWS * 64));
            flying = 8;
        }
        if constexpr (!(ABL & 1)) fb_dw_step<2, ABL, DPF>(acc_l1, bs_xyz[0], lds, a_off64, slot_ptr(bslot) + b_off64);    // delta^T @ xyz encoding
        bslot = next_slot(bslot);
        fb_wait_barrier<ABL>(flying);
    }

    // ---- this workgroup's partial.  Tile (qa, qb), lane (g, col), register r: dW[4 (4 g + r) + qa][4 col + qb] (32-row products:
    //      row 2 (4 g + r) + qa2); bias sums: fold the 4 lane groups (samples = g mod 4), feature 4 col + qa (2 col + qa2)
    float* out = fa.partial + (int64_t)blockIdx.x * FB_PART;
    auto store64 = [&](float* p, const f32x4 (&a)[2]) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            typedef float f32x2a __attribute__((ext_vector_type(2), aligned(8)));
            *reinterpret_cast<f32x2a*>(p + (4 * (4 * g + r) + qa) * 64 + 4 * col + qb0) = f32x2a{a[0][r], a[1][r]};
        }
    };
    auto store32 = [&](float* p, const f32x4 (&a)[1]) {
#pragma unroll
        for (int r = 0; r < 4; ++r) p[(2 * (4 * g + r) + qa2) * 64 + 4 * col + qb2] = a[0][r];
    };
    auto fold = [&](float v) {
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        return v;
    };
    store32(out + FB_P_DIRF, acc_dirf);
    store32(out + FB_P_DIRE, acc_dire);
    store64(out + FB_P_FEAT, acc_feat);
#pragma unroll
    for (int i = 1; i < MAXL - 1; ++i)          // (layers_xyz[0]'s comes out of the 64 x 64 sums below)
        if (i <= L - 2) store64(out + FB_P_XYZ + i * 4096, acc_xyz[i]);
    if (sk >= 0) store64(out + FB_P_SKIP, acc_skip);
    // S = [delta^T @ enc | column sums of delta] of the last delta -- tile-distributed in acc_l1, the sums in bs_xyz[0] -- goes through
    // LDS as a 64 x 64 matrix: columns < dx the product, column 63 the sums (the encoding is at most 63 wide), the columns between
    // zero (they met the rows' unwritten padding).  Then, 32 MFMAs per wave each:
    //   grad(layer1) = layers_xyz[0]^T S: S in the chain's B-operand layout (a "sample" = a column), the transposed weights from the
    //     stream as for any chain stage;  column 63 of the result is layer1's bias gradient;
    //   grad(layers_xyz[0].weight) = S [W1 | b1]^T: S as the A operand, [W1 | b1]^T (fa.w1t: rows 0 .. dx-1 and dx) as B.
    {
        stream_to_lds<8>(args.wstream + (2 * L - 1) * FB_CHUNK, lds, 2 * FB_CHUNK, wave, lane);      // layers_xyz[0]^T, both chunks
        float* mt = reinterpret_cast<float*>(lds + FB_OFF_DBUF);                                     // [k][64]
        const int dx = fa.dx;
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int j = 4 * col + qb0 + t;
                mt[(4 * (4 * g + r) + qa) * 64 + j] = j < dx ? acc_l1[t][r] : 0.0f;
            }
        fb_wait_barrier<ABL>(0);
        const float colsum = fold(bs_xyz[0]);
        if (bias64 && g == 0) mt[(4 * col + qa) * 64 + 63] = colsum;
        fb_wait_barrier<ABL>(0);
        const int jt = wave & 3;
        const bool upper = (wave & 4) != 0;                       // row tiles 2, 3 (else 0, 1)
        f32x4 o2[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const float b = mt[(16 * (ks >> 2) + 4 * g + (ks & 3)) * 64 + 16 * jt + col];
            const f32x4 a = *reinterpret_cast<const f32x4*>(lds + ks * 1024 + lane * 16);
            o2[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(upper ? a[2] : a[0], b, o2[0], 0, 0, 0);
            o2[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(upper ? a[3] : a[1], b, o2[1], 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * ((upper ? 2 : 0) + t) + 4 * g + r;
                out[FB_P_L1 + row * 64 + 16 * jt + col] = o2[t][r];
                if (jt == 3 && col == 15) out[FB_P_BIAS + FB_B_L1 + row] = o2[t][r];
            }
        // grad(layers_xyz[0].weight)[o][i] = sum_j S[o][j] Wx[j][i]:  this wave's column tile jt (i = 16 jt + col), row tiles as above
        f32x4 o3[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const int j = 4 * ks + g;                                                      // contraction index of this lane group
            const int wrow = j < dx ? j : dx;                                              // (column 63 of S pairs with the bias row)
            const float wv = fa.w1t[wrow * 64 + 16 * jt + col];
            const float b = (j < dx || j == 63) ? wv : 0.0f;
            const int r0 = 16 * (upper ? 2 : 0) + col;
            o3[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(mt[r0 * 64 + j], b, o3[0], 0, 0, 0);
            o3[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(mt[(r0 + 16) * 64 + j], b, o3[1], 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                out[FB_P_XYZ + (16 * ((upper ? 2 : 0) + t) + 4 * g + r) * 64 + 16 * jt + col] = o3[t][r];
    }
    float* ob = out + FB_P_BIAS;
    {
        const float v = fold(bs_dir);
        if (bias32 && g == 0) ob[FB_B_DIR + 2 * col + qa2] = v;
    }
    {
        const float v = fold(bs_feat);
        if (bias64 && g == 0) ob[FB_B_FEAT + 4 * col + qa] = v;
    }
#pragma unroll
    for (int i = 0; i < MAXL - 1; ++i) {
        const float v = fold(bs_xyz[i]);
        if (i <= L - 2 && bias64 && g == 0) ob[FB_B_XYZ + i * 64 + 4 * col + qa] = v;
    }
    // the heads: D rows 0..3 (the live rows of the delta tile) are registers 0..3 of lane group 0.  The waves that split a tile's
    // k-groups between them add their accumulators up through LDS, in k-part order (deterministic): one partial per workgroup
    float* oh = out + FB_P_HEAD;
    float* red = reinterpret_cast<float*>(lds);          // [4 k-parts][2 tiles][4][16] | [2 k-parts][4 tiles][4][16] | [2][4]
    fb_wait_barrier<ABL>(0);
    {
        const float v = fold(bs_head);
        if (g == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                red[((hr_k * 2 + (hr_t - 2)) * 4 + r) * 16 + col] = acc_rgb[r];
                red[512 + ((ha_k * 4 + ha_t) * 4 + r) * 16 + col] = acc_alpha[r];
            }
            if (ha_t == 0 && col < 4) red[1024 + ha_k * 4 + col] = v;
        }
    }
    fb_wait_barrier<ABL>(0);
    const int e = threadIdx.x;
    if (e < 128) oh[FB_H_RGB + e] = ((red[e] + red[128 + e]) + red[256 + e]) + red[384 + e];
    if (e < 256) oh[FB_H_ALPHA + e] = red[512 + e] + red[768 + e];
    if (e < 4) oh[FB_H_BIAS + e] = red[1024 + e] + red[1028 + e];
}

// out[o * out_ld + col0 + c] = sum over the parts p (index order) and a job's k-parts of
//     partial[p][off + (c >> 4) * tile_stride + o * row_stride + (c & 15) + k * kstride],   c < cols;   blockIdx.y = job.
// The layers' products are rows of 64 floats (row_stride 64, tile_stride 16, one k-part), the heads' [k-part][tile][row][16].
// The jobs behind `first_bias` are vectors: `rows` entries at partial[p][off + o + k * kstride].
struct FbReduceJob { int32_t off, rows, cols, out_ld, out_col0, row_stride, tile_stride, ksplit, kstride; float* out; };
constexpr int FB_MAX_JOBS = 2 * (FB_MAXL + 5);
struct FbReduce { FbReduceJob job[FB_MAX_JOBS]; int32_t first_bias; };
__global__ __launch_bounds__(256) void fb_reduce_kernel(const float* __restrict__ partial, const FbReduce rb, const int parts) {
    // four lanes per output element, each adds a quarter of the parts (index order, 16 loads in flight), then the four sums are
    // added in lane order: a fixed grouping -- deterministic -- whose dependent chain is a quarter as long (one thread per element
    // took 17 us behind a 260 us kernel: 256 dependent additions at HBM latency per batch of 16)
    const FbReduceJob j = rb.job[blockIdx.y];
    const int t = blockIdx.x * 64 + (threadIdx.x >> 2), sub = threadIdx.x & 3;
    const bool is_bias = (int)blockIdx.y >= rb.first_bias;
    const int elems = is_bias ? j.rows : j.rows * j.cols;
    const bool live = t < elems;
    const int tt = live ? t : 0;
    const int o = is_bias ? tt : tt / j.cols, c = is_bias ? 0 : tt - o * j.cols;
    const float* p = partial + j.off + (is_bias ? o : (c >> 4) * j.tile_stride + o * j.row_stride + (c & 15));
    const int total = parts * j.ksplit;
    const int per = (total + 3) / 4;
    const int e0 = sub * per, e1 = e0 + per < total ? e0 + per : total;
    auto addend = [&](int e) -> const float* {
        const int part = j.ksplit == 1 ? e : e / j.ksplit;
        return p + (int64_t)part * FB_PART + (e - part * j.ksplit) * j.kstride;
    };
    float s = 0.0f;
    int e = e0;
    for (; e + 16 <= e1; e += 16) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = *addend(e + u);
#pragma unroll
        for (int u = 0; u < 16; ++u) s += v[u];
    }
    for (; e < e1; ++e) s += *addend(e);
    const float s1 = __shfl_down(s, 1), s2 = __shfl_down(s, 2), s3 = __shfl_down(s, 3);
    if (live && sub == 0) j.out[is_bias ? o : (int64_t)o * j.out_ld + j.out_col0 + c] = ((s + s1) + s2) + s3;
}

#ifndef NM_FB_KERNEL_ONLY
static int fb_skip_layer(const nm_mlp* m, bool* ok) {
    // the one trunk layer that takes cat(x, xyz) (bit i of skip_mask, i <= L - 2); *ok = 0 when there are several
    int sk = -1, count = 0;
    for (int i = 0; i <= m->desc.num_layers - 2; ++i)
        if ((m->base.skip_mask >> i) & 1u) { sk = i; ++count; }
    *ok = count <= 1;
    return sk;
}

}  // namespace nm

using namespace nm;

extern "C" {

int nm_mlp_backward_fused_supported(const nm_mlp* m, int64_t n) {
    if (!m || m->lw || m->plan->generic_nt != 0 || m->precision != NM_PREC_F32) return 0;
    if (const char* v = getenv("NM_FUSED_BACKWARD"))           // A/B hook of the tools and tests (read per call)
        if (atoi(v) == 0) return 0;
    const nm_mlp_desc& d = m->desc;
    bool one_skip = false;
    fb_skip_layer(m, &one_skip);
    return d.hidden_size == 64 && d.use_viewdirs && d.num_layers >= 2 && d.num_layers <= FB_MAXL && one_skip && n > 0 &&
           n % FB_ROWS == 0 && nm_mlp_tapes_encodings(m) && 6 * d.num_encoding_fn_dir + (d.include_input_dir ? 3 : 0) <= 32;
}

int64_t nm_mlp_backward_fused_workspace_bytes(const nm_mlp* m) {      // the workgroups' partials, then [layer1.weight | bias]^T (64 x 64)
    return m ? (int64_t)(m->num_cus > 0 ? m->num_cus : 256) * FB_PART * 4 + 64 * 64 * 4 : 0;
}

int nm_mlp_backward_fused(nm_mlp* m, int64_t n, const nm_mlp_tape* tape, const float* d_radiance, const float* d_grad_radiance,
                          float* d_last, const nm_mlp_param_grads* grads, void* d_workspace, void* stream_) {
    NM_REQUIRE(m && tape && d_radiance && d_grad_radiance && grads && d_workspace, "bad argument");
    NM_REQUIRE(nm_mlp_backward_fused_supported(m, n), "this handle / sample count is not served by the fused backward (ask nm_mlp_backward_fused_supported)");
    NM_REQUIRE(tape->d_h && tape->d_feat && tape->d_mask_h && tape->d_mask_v && tape->d_enc_xyz && tape->d_enc_dir, "incomplete tape");
    NM_REQUIRE(tape->d_v == tape->d_enc_dir + 32 && tape->v_stride == 64,
               "the fused backward reads the view layer's activation rows out of the direction-encoding rows: tape the forward with d_v = d_enc_dir + 32, v_stride = 64");
    const nm_mlp_desc& d = m->desc;
    const int L = d.num_layers;
    NM_REQUIRE(grads->layer1_weight && grads->layer1_bias && grads->feat_weight && grads->feat_bias && grads->dir_weight && grads->dir_bias &&
               grads->alpha_weight && grads->alpha_bias && grads->rgb_weight && grads->rgb_bias, "incomplete gradient buffers");
    for (int i = 0; i <= L - 2; ++i) NM_REQUIRE(grads->xyz_weight[i] && grads->xyz_bias[i], "incomplete gradient buffers");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    MlpBwdArgs a = m->bwd;
    a.radiance = d_radiance; a.grad_out = d_grad_radiance;
    a.mask_h = tape->d_mask_h; a.mask_v = tape->d_mask_v;
    a.n = n; a.tiles = (n + 15) / 16;
    a.d_last = d_last;
    bool one_skip = false;
    FusedBwdArgs fa;
    fa.tape_h = tape->d_h; fa.tape_feat = tape->d_feat; fa.enc_x = tape->d_enc_xyz; fa.enc_d = tape->d_enc_dir;
    fa.partial = static_cast<float*>(d_workspace);
    fa.skip_layer = fb_skip_layer(m, &one_skip);
    const int cus = m->num_cus > 0 ? m->num_cus : 256;
    // [W1 | b1]^T out of the packed image -- the operand of the epilogue's product for layers_xyz[0] -- behind the partials
    float* w1t = fa.partial + (int64_t)cus * FB_PART;
    if (int rc = nm_mlp_export_layer1_transposed(m, w1t, stream_)) return rc;
    fa.w1t = w1t;
    fa.dx = 6 * d.num_encoding_fn_xyz + (d.include_input_xyz ? 3 : 0);
    const int64_t wg_iters = n / FB_ROWS;
    const int grid = (int)(wg_iters < cus ? wg_iters : cus);
    const auto kernel = L <= 4 ? &mlp_backward_dw64_kernel<4> : &mlp_backward_dw64_kernel<FB_MAXL>;
    if (int rc = ensure_dynamic_lds((const void*)kernel, FB_LDS)) return rc;
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(512), FB_LDS, stream, a, fa, L);
    // ---- the order-fixed reduction of the per-workgroup partials into the gradient tensors
    const int dx = 6 * d.num_encoding_fn_xyz + (d.include_input_xyz ? 3 : 0), dd = 6 * d.num_encoding_fn_dir + (d.include_input_dir ? 3 : 0);
    FbReduce rb;
    int nj = 0;
    auto job = [&](int off, int rows, int cols, int ld, int col0, float* out) { rb.job[nj++] = FbReduceJob{off, rows, cols, ld, col0, 64, 16, 1, 0, out}; };
    job(FB_P_DIRF, 32, 64, 64 + dd, 0, grads->dir_weight);
    if (dd > 0) job(FB_P_DIRE, 32, dd, 64 + dd, 64, grads->dir_weight);
    job(FB_P_FEAT, 64, 64, 64, 0, grads->feat_weight);
    for (int i = 0; i <= L - 2; ++i) {
        const bool skip = i == fa.skip_layer;
        job(FB_P_XYZ + i * 4096, 64, 64, skip ? 64 + dx : 64, 0, grads->xyz_weight[i]);
        if (skip) job(FB_P_SKIP, 64, dx, 64 + dx, 64, grads->xyz_weight[i]);
    }
    job(FB_P_L1, 64, dx, dx, 0, grads->layer1_weight);
    rb.job[nj++] = FbReduceJob{FB_P_HEAD + FB_H_RGB, 3, 32, 32, 0, 16, 64, 1, 0, grads->rgb_weight};            // rows 0..2 of d_last^T @ v
    rb.job[nj++] = FbReduceJob{FB_P_HEAD + FB_H_ALPHA + 3 * 16, 1, 64, 64, 0, 16, 64, 1, 0, grads->alpha_weight};   // row 3 of d_last^T @ h[L-1]
    rb.first_bias = nj;
    job(FB_P_BIAS + FB_B_DIR, 32, 1, 1, 0, grads->dir_bias);
    job(FB_P_BIAS + FB_B_FEAT, 64, 1, 1, 0, grads->feat_bias);
    for (int i = 0; i <= L - 2; ++i) job(FB_P_BIAS + FB_B_XYZ + i * 64, 64, 1, 1, 0, grads->xyz_bias[i]);
    job(FB_P_BIAS + FB_B_L1, 64, 1, 1, 0, grads->layer1_bias);
    rb.job[nj++] = FbReduceJob{FB_P_HEAD + FB_H_BIAS, 3, 1, 1, 0, 0, 0, 1, 0, grads->rgb_bias};
    rb.job[nj++] = FbReduceJob{FB_P_HEAD + FB_H_BIAS + 3, 1, 1, 1, 0, 0, 0, 1, 0, grads->alpha_bias};
    hipLaunchKernelGGL(fb_reduce_kernel, dim3(64, nj), dim3(256), 0, stream, fa.partial, rb, grid);
    NM_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // extern "C"
#else
}  // namespace nm
#endif  // NM_FB_KERNEL_ONLY (tests/tools/probes/fb_probe.hip includes the kernels alone)
