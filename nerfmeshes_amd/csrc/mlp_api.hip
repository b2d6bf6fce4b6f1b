// C ABI for the fused MLP: weight packing (torch.nn.Linear layout -> MFMA A-operand stream),
// handle lifetime, and the three entry points that launch nerf_mlp.hip's kernel.
#include <array>
#include <cstring>
#include <map>
#include <mutex>
#include <utility>
#include <vector>

#include "nm_internal.h"
#include "mlp_device_g.h"
#include "nerf_layerwise.h"

namespace nm {

static thread_local std::string g_error;
void set_error(const std::string& msg) { g_error = msg; }

int ensure_dynamic_lds(const void* kernel, int bytes) {
    NM_REQUIRE(bytes <= 160 * 1024, "LDS budget exceeded");
    static std::mutex lock;
    static std::map<std::pair<int, const void*>, int> have;
    int dev = 0;
    NM_HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> guard(lock);
    int& cur = have[{dev, kernel}];
    if (cur < bytes) {
        NM_HIP_CHECK(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        cur = bytes;
    }
    return 0;
}

const MlpPlan* find_mlp_plan(int H, int FX, int FD);
const MlpPlan* find_generic_plan(int H, int L, bool long_encoding);
bool has_b3_kernel(int H, int FX, int FD);
int mlp_plan_info(const MlpPlan* p, int* nw);
int launch_mlp(const nm_mlp* m, const MlpArgs& args, int density_only, hipStream_t stream);

// Source column (input feature) of the weight matrix that lane group g consumes at k-step s.
using StepCols = std::array<int, 4>;  // -1 = zero padding

// hidden activation of width H produced by MFMA tiles: k-step s = 4*tile + reg, group g holds
// feature 16*tile + 4*g + reg (see nerf_mlp.hip header).
static void hidden_steps(std::vector<StepCols>& out, int width, int col_offset) {
    for (int s = 0; s < width / 4; ++s) {
        StepCols c;
        for (int g = 0; g < 4; ++g) c[g] = col_offset + 16 * (s >> 2) + 4 * g + (s & 3);
        out.push_back(c);
    }
}

// positional encoding [x(3) | sin(3F) | cos(3F)], coordinate-major (modules.py:26-34): k-step s
// carries arguments a0=2s (groups 0,1 = sin,cos) and a1=2s+1 (groups 2,3); last step = identity.
static void encoding_steps(std::vector<StepCols>& out, int F, bool include_input, int col_offset) {
    const int base = col_offset + (include_input ? 3 : 0);
    for (int s = 0; s < (3 * F + 1) / 2; ++s) {
        StepCols c;
        for (int g = 0; g < 4; ++g) {
            const int a = 2 * s + (g >> 1);
            c[g] = a < 3 * F ? base + ((g & 1) ? 3 * F : 0) + a : -1;
        }
        out.push_back(c);
    }
    StepCols id;
    for (int g = 0; g < 4; ++g) id[g] = (include_input && g < 3) ? col_offset + g : -1;
    out.push_back(id);
}

// Append the A-operand stream of one GEMM as SOURCE INDICES (tensor id << 24 | element): for k-step s,
// block b of VW tiles, lane l, slot q: W[16*(VW*b+q) + (l&15)][cols[s][l>>4]]; `transposed` addresses W^T
// (the backward stream: output row n is an input column of the stored nn.Linear weight).
static void pack_gemm(std::vector<int32_t>& out, int tensor, int ld, int rows, int ntiles,
                      const std::vector<StepCols>& steps, bool transposed = false) {
    const int vw = ntiles >= 4 ? 4 : ntiles;
    for (const StepCols& c : steps)
        for (int b = 0; b < ntiles / vw; ++b)
            for (int l = 0; l < 64; ++l)
                for (int q = 0; q < vw; ++q) {
                    const int n = 16 * (vw * b + q) + (l & 15);
                    const int k = c[l >> 4];
                    const int64_t off = transposed ? (int64_t)k * ld + n : (int64_t)n * ld + k;
                    out.push_back((n < rows && k >= 0) ? (int32_t)((tensor << 24) | (int32_t)off) : -1);
                }
}

static bool is_skip(const nm_mlp_desc& d, int i);
static void pack_range(std::vector<int32_t>& out, int tensor, int count);
static void pad_to(std::vector<int32_t>& v, size_t multiple);

// ---- generic-shape family (mlp_device_g.h) -------------------------------------------------------------------------
// hidden activation of the PADDED width 16 * nt, real width `width`: columns beyond it are zero weights
static void hidden_steps_g(std::vector<StepCols>& out, int nt, int width, int col_offset) {
    for (int s = 0; s < 4 * nt; ++s) {
        StepCols c;
        for (int g = 0; g < 4; ++g) {
            const int k = 16 * (s >> 2) + 4 * g + (s & 3);
            c[g] = k < width ? col_offset + k : -1;
        }
        out.push_back(c);
    }
}

// an encoding's own stage: (3 F + 1) / 2 argument k-steps, the identity step only when the input is included, zero k-steps
// up to a whole number of chunks; returns the chunk count
static int encoding_stage_g(std::vector<StepCols>& out, int F, bool include_input, int col_offset, int kch) {
    const int base = col_offset + (include_input ? 3 : 0);
    int n = 0;
    for (int s = 0; s < (3 * F + 1) / 2; ++s, ++n) {
        StepCols c;
        for (int g = 0; g < 4; ++g) {
            const int a = 2 * s + (g >> 1);
            c[g] = a < 3 * F ? base + ((g & 1) ? 3 * F : 0) + a : -1;
        }
        out.push_back(c);
    }
    if (include_input) {
        StepCols id;
        for (int g = 0; g < 4; ++g) id[g] = g < 3 ? col_offset + g : -1;
        out.push_back(id);
        ++n;
    }
    for (; n % kch; ++n) out.push_back(StepCols{-1, -1, -1, -1});
    return n / kch;
}

// A-operand stream of one generic stage: per k-step ceil(ntiles / 4) blocks of 4 tiles (1 KiB each; tiles beyond ntiles and rows
// beyond `rows` are zeros)
static void pack_gemm_g(std::vector<int32_t>& out, int tensor, int ld, int rows, int ntiles, const std::vector<StepCols>& steps,
                        bool transposed = false) {
    const int nb = (ntiles + 3) / 4;
    for (const StepCols& c : steps)
        for (int b = 0; b < nb; ++b)
            for (int l = 0; l < 64; ++l)
                for (int q = 0; q < 4; ++q) {
                    const int n = 16 * (4 * b + q) + (l & 15);
                    const int k = c[l >> 4];
                    const int64_t off = transposed ? (int64_t)k * ld + n : (int64_t)n * ld + k;    // transposed: W^T (the delta stream)
                    out.push_back((4 * b + q < ntiles && n < rows && k >= 0) ? (int32_t)((tensor << 24) | (int32_t)off) : -1);
                }
}

static void pack_range_padded(std::vector<int32_t>& out, int tensor, int count, int padded) {
    for (int i = 0; i < padded; ++i) out.push_back(i < count ? ((tensor << 24) | i) : -1);
}

// GEMV operand of a head row over a D-layout activation of padded width 16 * nt (fc_alpha's layout): [4 lane groups][4 nt]
static void pack_head_row_g(std::vector<int32_t>& out, int tensor, int row_offset, int nt, int width) {
    for (int g = 0; g < 4; ++g)
        for (int s = 0; s < 4 * nt; ++s) {
            const int k = 16 * (s >> 2) + 4 * g + (s & 3);
            out.push_back(k < width ? ((tensor << 24) | (row_offset + k)) : -1);
        }
}

struct BlobLayout { size_t off_bias, off_wa, off_wr, off_bwd; uint32_t skip_mask; int chx, chd; };

// ---- layer-wise path (nerf_layerwise.hip): networks beyond the fused families' limits -----------------------------------
// A plan that launches nothing by itself: launch_mlp_timed / the training entry points route such a handle to the layer-wise
// evaluator.  nm_mlp_kernel_variant reports 2000.
static const MlpPlan g_layerwise_plan = {0, 0, 0, 8, 0, 2000, 0, false, nullptr, 0, 0, nullptr, 0, nullptr, nullptr};

// The blob of a layer-wise handle as an index map: per Linear its transpose (in x out: the forward products' A operand), the
// matrix itself (out x in: the delta chain's), its bias; every piece 256-byte aligned (16-byte DMA pieces need it).
static void build_index_layerwise(std::vector<int32_t>& index, const nm_mlp_desc& d, LwNet* net) {
    const int H = d.hidden_size, L = d.num_layers, FX = d.num_encoding_fn_xyz, FD = d.num_encoding_fn_dir;
    const bool no_view = d.use_viewdirs == 0;
    const int dx = 6 * FX + (d.include_input_xyz ? 3 : 0), dd = no_view ? 0 : 6 * FD + (d.include_input_dir ? 3 : 0);
    auto linear = [&](int tw, int tb, int out, int in, int nbias) {
        LwLinear l;
        l.out = out; l.in = in;
        pad_to(index, 64); l.wt = index.size();
        for (int k = 0; k < in; ++k)
            for (int o = 0; o < out; ++o) index.push_back((tw << 24) | (o * in + k));
        pad_to(index, 64); l.w = index.size();
        pack_range(index, tw, out * in);
        pad_to(index, 64); l.b = index.size();
        pack_range(index, tb, nbias);
        return l;
    };
    net->L = L; net->H = H; net->H2 = H / 2; net->dx = dx; net->dd = dd; net->flat = no_view ? 1 : 0;
    net->fx = FX; net->fd = no_view ? 0 : FD; net->inc_x = d.include_input_xyz ? 1 : 0; net->inc_d = d.include_input_dir ? 1 : 0;
    net->skip_mask = 0;
    net->layer1 = linear(T_L1W, T_L1B, H, dx, H);
    for (int i = 0; i < L - 1; ++i) {
        const bool skip = is_skip(d, i);
        if (skip) net->skip_mask |= 1u << i;
        net->xyz[i] = linear(T_XYZ0 + 2 * i, T_XYZ0 + 2 * i + 1, H, H + (skip ? dx : 0), H);
    }
    net->alpha = linear(T_ALPHAW, T_ALPHAB, 1, H, 1);
    if (no_view) {
        net->rgb = linear(T_RGBW, T_RGBB, 3, H, 3);              // rows 0..2 of fc_out (nm_mlp_weights)
    } else {
        net->feat = linear(T_FEATW, T_FEATB, H, H, H);
        net->dir = linear(T_DIRW, T_DIRB, H / 2, H + dd, H / 2);
        net->rgb = linear(T_RGBW, T_RGBB, 3, H / 2, 3);
    }
    index.resize(index.size() + 1024, -1);                       // what a piece's rounding may read past the last matrix
}

// The whole blob of a generic plan as an index map (layout: mlp_device_g.h's kernels): forward stream | biases | fc_alpha |
// fc_rgb (or fc_out's colour rows) | backward stream (the transposed layers in reverse order, hidden columns only).
static BlobLayout build_index_generic(std::vector<int32_t>& index, const nm_mlp_desc& d, const MlpPlan& plan) {
    const int H = d.hidden_size, L = d.num_layers, FX = d.num_encoding_fn_xyz, FD = d.num_encoding_fn_dir;
    const bool no_view = d.use_viewdirs == 0;
    const int dx = 6 * FX + (d.include_input_xyz ? 3 : 0), dd = 6 * FD + (d.include_input_dir ? 3 : 0);
    const int NT = plan.generic_nt, NTD = (NT + 1) / 2, HP = 16 * NT, HPD = 16 * NTD, kch = plan.KCH;
    BlobLayout lay{};
    std::vector<StepCols> enc_x, hid, skip_enc, dir_enc;
    lay.chx = encoding_stage_g(enc_x, FX, d.include_input_xyz != 0, 0, kch);
    hidden_steps_g(hid, NT, H, 0);
    encoding_stage_g(skip_enc, FX, d.include_input_xyz != 0, H, kch);         // cat(hidden, xyz): models.py:65
    lay.chd = no_view ? 0 : encoding_stage_g(dir_enc, FD, d.include_input_dir != 0, H, kch);   // cat(feat, view): models.py:72
    pack_gemm_g(index, T_L1W, dx, H, NT, enc_x);
    for (int i = 0; i < L - 1; ++i) {
        const bool skip = is_skip(d, i);
        const int ld = H + (skip ? dx : 0);
        pack_gemm_g(index, T_XYZ0 + 2 * i, ld, H, NT, hid);
        if (skip) {
            pack_gemm_g(index, T_XYZ0 + 2 * i, ld, H, NT, skip_enc);
            lay.skip_mask |= 1u << i;
        }
    }
    if (!no_view) {
        pack_gemm_g(index, T_FEATW, H, H, NT, hid);
        pack_gemm_g(index, T_DIRW, H + dd, H / 2, NTD, hid);
        pack_gemm_g(index, T_DIRW, H + dd, H / 2, NTD, dir_enc);
    }
    index.resize(index.size() + 8192, -1);   // DMA granularity padding: a tail fetch may run one chunk (32 KiB) past the stream
    pad_to(index, 64);
    lay.off_bias = index.size();
    pack_range_padded(index, T_L1B, H, HP);
    for (int i = 0; i < L - 1; ++i) pack_range_padded(index, T_XYZ0 + 2 * i + 1, H, HP);
    if (no_view) index.resize(index.size() + HP + HPD, -1);
    else {
        pack_range_padded(index, T_FEATB, H, HP);
        pack_range_padded(index, T_DIRB, H / 2, HPD);
    }
    pack_range(index, T_ALPHAB, 1);
    pack_range(index, T_RGBB, 3);
    pad_to(index, 64);
    lay.off_wa = index.size();
    pack_head_row_g(index, T_ALPHAW, 0, NT, H);
    pad_to(index, 64);
    lay.off_wr = index.size();
    for (int c = 0; c < 3; ++c) {
        if (no_view) pack_head_row_g(index, T_RGBW, c * H, NT, H);              // rows 0..2 of fc_out over the trunk output
        else pack_head_row_g(index, T_RGBW, c * (H / 2), NTD, H / 2);
    }
    pad_to(index, 64);
    lay.off_bwd = index.size();
    std::vector<StepCols> hid_half;
    hidden_steps_g(hid_half, NTD, H / 2, 0);
    if (!no_view) {
        pack_gemm_g(index, T_DIRW, H + dd, H, NT, hid_half, true);     // delta_v (H/2 columns) -> delta at relu(fc_feat) (H rows)
        pack_gemm_g(index, T_FEATW, H, H, NT, hid, true);
    }
    for (int i = L - 2; i >= 0; --i) pack_gemm_g(index, T_XYZ0 + 2 * i, H + (is_skip(d, i) ? dx : 0), H, NT, hid, true);
    index.resize(index.size() + 8192, -1);
    pad_to(index, 64);
    return lay;
}

// the per-argument table of one encoding (GEncArg: band, coordinate)
static void fill_enc_table(float* tab /* [parts * G_ENC_ARGS][2] */, int parts, int F, const float* bands) {
    for (int a = 0; a < parts * G_ENC_ARGS; ++a) {
        const bool real = F > 0 && a < 3 * F;
        const int32_t coord = real ? a / F : 0;
        tab[2 * a] = real ? bands[a % F] : 0.0f;
        memcpy(&tab[2 * a + 1], &coord, 4);
    }
}

// ---- bf16x3 stream (mlp_device_b3.h): per (k-block m, tile nt) unit the fp32 image [lane][j = 0..7] =
// W[16 nt + (l & 15)][column of slot (m, l >> 4, j)]; a device kernel splits it into the three bf16 planes.
using SlotCols = std::array<int, 32>;   // source column of slot 8 g + j of one k-block (-1 = zero)

static void hidden_blocks(std::vector<SlotCols>& out, int width, int col_offset) {
    for (int m = 0; m < width / 32; ++m) {
        SlotCols c;
        for (int g = 0; g < 4; ++g)
            for (int j = 0; j < 8; ++j) c[8 * g + j] = col_offset + 16 * (2 * m + j / 4) + 4 * g + (j % 4);
        out.push_back(c);
    }
}

// slots 2a, 2a+1 = sin, cos of argument a < 3F (reference columns base + a, base + 3F + a); then the identity coordinates
static void encoding_blocks(std::vector<SlotCols>& out, int F, bool include_input, int col_offset, int blocks) {
    const int base = col_offset + (include_input ? 3 : 0);
    for (int m = 0; m < blocks; ++m) {
        SlotCols c;
        for (int q = 0; q < 32; ++q) {
            const int s = 32 * m + q;
            if (s < 6 * F) c[q] = base + ((s & 1) ? 3 * F : 0) + s / 2;
            else if (include_input && s - 6 * F < 3) c[q] = col_offset + (s - 6 * F);
            else c[q] = -1;
        }
        out.push_back(c);
    }
}

static void pack_gemm_b3(std::vector<int32_t>& out, int tensor, int ld, int rows, int ntiles, const std::vector<SlotCols>& blocks) {
    for (const SlotCols& c : blocks)
        for (int nt = 0; nt < ntiles; ++nt)
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < 8; ++j) {
                    const int n = 16 * nt + (l & 15), k = c[8 * (l >> 4) + j];
                    out.push_back((n < rows && k >= 0) ? (int32_t)((tensor << 24) | (int32_t)((int64_t)n * ld + k)) : -1);
                }
}

// fp32 image [unit][lane][8] -> three planes of packed bf16 [unit][plane][lane][8]: x1 = bf16(x), x2 = bf16(x - x1), ...
__global__ void split_b3_kernel(const float* __restrict__ src, uint4* __restrict__ dst, int64_t units) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;     // (unit, lane)
    if (t >= units * 64) return;
    const int64_t unit = t >> 6;
    const int lane = (int)(t & 63);
    const float* v = src + t * 8;
    uint32_t p[3][4];
    for (int q = 0; q < 4; ++q) {
        const float x = v[2 * q], y = v[2 * q + 1];
        auto pack = [](float a, float b) {
            typedef float f2 __attribute__((ext_vector_type(2)));
            typedef __bf16 b2 __attribute__((ext_vector_type(2)));
            const f2 f = {a, b};
            return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, b2));
        };
        p[0][q] = pack(x, y);
        const float rx = x - __uint_as_float(p[0][q] << 16), ry = y - __uint_as_float(p[0][q] & 0xffff0000u);
        p[1][q] = pack(rx, ry);
        const float sx = rx - __uint_as_float(p[1][q] << 16), sy = ry - __uint_as_float(p[1][q] & 0xffff0000u);
        p[2][q] = pack(sx, sy);
    }
    for (int k = 0; k < 3; ++k) dst[(unit * 3 + k) * 64 + lane] = uint4{p[k][0], p[k][1], p[k][2], p[k][3]};
}

static void pack_range(std::vector<int32_t>& out, int tensor, int count) {
    for (int i = 0; i < count; ++i) out.push_back((tensor << 24) | i);
}

static void pad_to(std::vector<int32_t>& v, size_t multiple) {
    while (v.size() % multiple) v.push_back(-1);
}

// blob[i] = parameter element index[i] names (or 0): the whole packing, on the device.  Every pass also folds what it read into
// a 64-bit checksum of the packed image (sum over i of bits(value_i) * (2 i + 1), integer arithmetic: order-free, so plain
// atomics keep it deterministic): WRITE = the gather itself, !WRITE = a verification pass over the caller's LIVE tensors that
// touches nothing (nm_mlp_weights_current: "is the packed copy still what these tensors hold?").
template <bool WRITE>
__global__ __launch_bounds__(256) void gather_parameters(const int32_t* __restrict__ index, WeightPtrs ptrs, float* __restrict__ blob,
                                                         int64_t count, unsigned long long* __restrict__ check) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long term = 0;
    if (i < count) {
        const int32_t s = index[i];
        const float v = s < 0 ? 0.0f : ptrs.p[s >> 24][s & 0xffffff];
        if (WRITE) blob[i] = v;
        term = (unsigned long long)__float_as_uint(v) * (unsigned long long)(2 * i + 1);
    }
    if (!check) return;
#pragma unroll
    for (int off = 32; off; off >>= 1) {
        const unsigned lo = __shfl_xor((unsigned)term, off), hi = __shfl_xor((unsigned)(term >> 32), off);
        term += ((unsigned long long)hi << 32) | lo;
    }
    __shared__ unsigned long long part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = term;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(check, (part[0] + part[1]) + (part[2] + part[3]));
}

// the inverse of the gather for one tensor: out[element] = blob[i] wherever index[i] names that tensor (a parameter element sits in
// the packed image at least once -- forward stream, transposed stream --, always with the same value)
__global__ __launch_bounds__(256) void scatter_tensor(const int32_t* __restrict__ index, const float* __restrict__ blob, int64_t count,
                                                      int tensor, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const int32_t s = index[i];
    if (s >= 0 && (s >> 24) == tensor) out[s & 0xffffff] = blob[i];
}

// [layer1.weight | layer1.bias]^T out of the packed image: out[(j, i)] = W1[i][j] for j < dx, out[(dx, i)] = b1[i]; (dx + 1, H) row-major
__global__ __launch_bounds__(256) void scatter_layer1_t(const int32_t* __restrict__ index, const float* __restrict__ blob, int64_t count,
                                                        int H, int dx, float* __restrict__ out) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= count) return;
    const int32_t s = index[p];
    if (s < 0) return;
    const int tensor = s >> 24, e = s & 0xffffff;
    if (tensor == T_L1W) out[(int64_t)(e % dx) * H + e / dx] = blob[p];
    else if (tensor == T_L1B) out[(int64_t)dx * H + e] = blob[p];
}

static bool is_skip(const nm_mlp_desc& d, int i) {   // models.py:37,63
    return i % d.skip_step == 0 && i > 0 && i != d.num_layers - 1;
}

static void weight_pointers(const nm_mlp_desc& d, const nm_mlp_weights& w, WeightPtrs& p) {
    std::memset(&p, 0, sizeof(p));
    p.p[T_L1W] = w.layer1_w; p.p[T_L1B] = w.layer1_b;
    for (int i = 0; i < d.num_layers - 1; ++i) {
        p.p[T_XYZ0 + 2 * i] = w.layers_xyz_w[i];
        p.p[T_XYZ0 + 2 * i + 1] = w.layers_xyz_b[i];
    }
    p.p[T_FEATW] = w.fc_feat_w; p.p[T_FEATB] = w.fc_feat_b;
    p.p[T_ALPHAW] = w.fc_alpha_w; p.p[T_ALPHAB] = w.fc_alpha_b;
    p.p[T_DIRW] = w.layers_dir0_w; p.p[T_DIRB] = w.layers_dir0_b;
    p.p[T_RGBW] = w.fc_rgb_w; p.p[T_RGBB] = w.fc_rgb_b;
}

static int launch_gather(nm_mlp* m, const WeightPtrs& ptrs, hipStream_t stream) {
    const int64_t n = (int64_t)m->blob_floats;
    if (m->d_check) NM_HIP_CHECK(hipMemsetAsync(m->d_check, 0, 8, stream));
    hipLaunchKernelGGL(gather_parameters<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, m->d_index, ptrs,
                       static_cast<float*>(m->d_blob), n, m->d_check);
    if (m->precision == NM_PREC_BF16X3) {
        const int64_t nb = (int64_t)m->b3_units * 512;
        hipLaunchKernelGGL(gather_parameters<true>, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, stream, m->d_index_b3, ptrs,
                           m->d_tmp_b3, nb, static_cast<unsigned long long*>(nullptr));
        hipLaunchKernelGGL(split_b3_kernel, dim3((unsigned)((m->b3_units * 64 + 255) / 256)), dim3(256), 0, stream,
                           m->d_tmp_b3, static_cast<uint4*>(m->d_stream_b3), (int64_t)m->b3_units);
    }
    NM_HIP_CHECK(hipGetLastError());
    ++m->refresh_count;
    return 0;
}

// ---- optional per-launch timing of the dominant kernel (bench.py's roofline leg) ---------------
struct ProfRec { hipEvent_t start, stop; double flops; };
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof;

// density_only: 0 full evaluation | 1 sigma only.  For a use_viewdirs = 0 handle the full evaluation is the kernels' mode 2:
// the trunk as in mode 1, then all four rows of fc_out (models.py:77-79).
static int launch_mlp_timed(const nm_mlp* m, const MlpArgs& a, int density_only, hipStream_t stream) {
    if (!density_only && !m->desc.use_viewdirs) density_only = 2;
    auto launch = [&]() { return m->lw ? layerwise_forward(const_cast<nm_mlp*>(m), a, density_only, stream) : launch_mlp(m, a, density_only, stream); };
    if (!g_prof_on || a.n <= 0) return launch();
    ProfRec r;
    NM_HIP_CHECK(hipEventCreate(&r.start));
    NM_HIP_CHECK(hipEventCreate(&r.stop));
    r.flops = (double)a.n * (double)(density_only == 1 ? m->flops_density : m->flops_full);
    NM_HIP_CHECK(hipEventRecord(r.start, stream));
    const int rc = launch();
    NM_HIP_CHECK(hipEventRecord(r.stop, stream));
    g_prof.push_back(r);
    return rc;
}

static int64_t mlp_macs(const nm_mlp_desc& d, bool density_only) {
    const int64_t H = d.hidden_size, dx = 6 * d.num_encoding_fn_xyz + (d.include_input_xyz ? 3 : 0);
    const int64_t dd = 6 * d.num_encoding_fn_dir + (d.include_input_dir ? 3 : 0);
    int64_t macs = dx * H;
    for (int i = 0; i < d.num_layers - 1; ++i) {
        const bool skip = i % d.skip_step == 0 && i > 0 && i != d.num_layers - 1;
        macs += (H + (skip ? dx : 0)) * H;
    }
    macs += H;  // fc_alpha (use_viewdirs = 0: row 3 of fc_out)
    if (!density_only) macs += d.use_viewdirs ? H * H + (H + dd) * (H / 2) + (H / 2) * 3 : 3 * H;   // else: rows 0..2 of fc_out
    return macs;
}

}  // namespace nm

using namespace nm;

extern "C" {

const char* nm_last_error(void) { return g_error.c_str(); }
int nm_abi_version(void) { return NM_ABI_VERSION; }
int nm_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return -1;
    return n;
}

int nm_mlp_profile_enable(int on) {
    g_prof_on = on != 0;
    return 0;
}

int nm_mlp_profile_read(int64_t* launches, double* total_ms, double* total_flops) {
    double ms = 0, fl = 0;
    for (ProfRec& r : g_prof) {
        NM_HIP_CHECK(hipEventSynchronize(r.stop));
        float t = 0;
        NM_HIP_CHECK(hipEventElapsedTime(&t, r.start, r.stop));
        ms += t; fl += r.flops;
        (void)hipEventDestroy(r.start); (void)hipEventDestroy(r.stop);
    }
    if (launches) *launches = (int64_t)g_prof.size();
    if (total_ms) *total_ms = ms;
    if (total_flops) *total_flops = fl;
    g_prof.clear();
    return 0;
}

int nm_mlp_create(const nm_mlp_desc* desc, const nm_mlp_weights* w, int device, nm_mlp** out) {
    return nm_mlp_create_ex(desc, w, device, NM_PREC_F32, out);
}

int nm_mlp_precision(const nm_mlp* m) { return m ? m->precision : -1; }

int nm_mlp_create_ex(const nm_mlp_desc* desc, const nm_mlp_weights* w, int device, int precision, nm_mlp** out) {
    NM_REQUIRE(desc && w && out, "null argument");
    const bool force_generic = (precision & NM_KERNEL_GENERIC) != 0;
    precision &= ~NM_KERNEL_GENERIC;
    NM_REQUIRE(precision == NM_PREC_F32 || precision == NM_PREC_BF16X3, "unknown precision");
    NM_REQUIRE(!force_generic || precision == NM_PREC_F32, "NM_KERNEL_GENERIC goes with NM_PREC_F32");
    const nm_mlp_desc& d = *desc;
    NM_REQUIRE(d.use_viewdirs == 0 || d.use_viewdirs == 1, "use_viewdirs is 0 or 1");
    const bool no_view = d.use_viewdirs == 0;      // models.py:77-79: trunk -> fc_out (4 rows), no view branch
    NM_REQUIRE(!no_view || precision == NM_PREC_F32, "use_viewdirs=0 networks run in fp32 only");
    NM_REQUIRE(d.num_layers >= 2 && d.num_layers <= 32, "num_layers out of range");
    NM_REQUIRE(d.skip_step >= 1, "skip_step must be >= 1");
    NM_REQUIRE(d.hidden_size >= 1 && d.num_encoding_fn_xyz >= 0 && d.num_encoding_fn_dir >= 0, "negative network dimension");
    NM_REQUIRE(no_view || d.hidden_size >= 2, "hidden_size = 1 with view directions: layers_dir[0] would have hidden_size // 2 = 0 rows");
    NM_REQUIRE(d.num_encoding_fn_xyz > 0 || d.include_input_xyz,
               "the xyz encoding is empty (num_encoding_fn_xyz = 0 without include_input_xyz): layer1 would have no input");
    // every tensor the packer will read, checked before anything dereferences one
    NM_REQUIRE(w->layer1_w && w->layer1_b && w->fc_alpha_w && w->fc_alpha_b && w->fc_rgb_w && w->fc_rgb_b, "missing weight tensor");
    NM_REQUIRE(no_view || (w->fc_feat_w && w->fc_feat_b && w->layers_dir0_w && w->layers_dir0_b), "missing view-branch weight tensor");
    NM_REQUIRE(w->layers_xyz_w && w->layers_xyz_b, "missing layers_xyz tensor tables");
    for (int i = 0; i < d.num_layers - 1; ++i)
        NM_REQUIRE(w->layers_xyz_w[i] && w->layers_xyz_b[i], "missing layers_xyz weight tensor");
    NM_REQUIRE(d.num_encoding_fn_xyz == 0 || w->freq_xyz, "missing xyz frequency bands");
    NM_REQUIRE(no_view || d.num_encoding_fn_dir == 0 || w->freq_dir, "missing direction frequency bands");
    const int H = d.hidden_size, L = d.num_layers, FX = d.num_encoding_fn_xyz, FD = d.num_encoding_fn_dir;
    const int dx = 6 * FX + (d.include_input_xyz ? 3 : 0), dd = 6 * FD + (d.include_input_dir ? 3 : 0);
    // A tuned plan for exactly this shape, else the generic family (mlp_device_g.h): every shape FlexibleNeRFModel's
    // constructor accepts up to hidden_size 512 and 24 k-steps per encoding.  (A network without view directions has no
    // direction encoding: any tuned kernel of that width / xyz encoding runs it.)
    const MlpPlan* plan = (!force_generic && FX <= MAX_FREQ_XYZ && (no_view || FD <= MAX_FREQ_DIR)) ? find_mlp_plan(H, FX, no_view ? 4 : FD) : nullptr;
    LwNet* lw_net = nullptr;
    if (!plan) {
        const int steps_x = (3 * FX + 1) / 2 + (d.include_input_xyz ? 1 : 0), steps_d = (3 * FD + 1) / 2 + (d.include_input_dir ? 1 : 0);
        const int steps = (!no_view && steps_d > steps_x) ? steps_d : steps_x;
        const bool long_encoding = steps > G_ENC_PARTS * G_ENC_STEPS;      // beyond what the fused kernels take even in two parts
        plan = find_generic_plan(H, L, steps > G_ENC_STEPS);
        if (precision != NM_PREC_F32) {
            set_error("precision bf16x3 is instantiated for hidden_size 64 / 128 / 256 with 6 or 10 xyz / 4 direction frequencies only");
            return 3;
        }
        if (!plan || long_encoding) {
            // beyond the fused families -- hidden_size > 512 (half of a wider layer's activations does not fit the register file of
            // one wavefront), an encoding of more than 48 MFMA k-steps (31 functions), or so many layers that their biases no longer
            // fit the LDS next to the weight ring --: the layer-wise path (nerf_layerwise.hip)
            if (FX > LW_MAX_FREQ || (!no_view && FD > LW_MAX_FREQ)) {
                set_error("an encoding of " + std::to_string(FX) + " / " + std::to_string(FD) + " functions: the limit is " +
                          std::to_string(LW_MAX_FREQ) + " per input (frequency 2^31 is past fp32's integer range)");
                return 3;
            }
            if ((int64_t)H * (H + dx) >= (1 << 24)) {
                set_error("hidden_size=" + std::to_string(H) + ": a weight matrix of more than 2^24 elements exceeds the packer's index map");
                return 3;
            }
            plan = &g_layerwise_plan;
            lw_net = new LwNet();
            std::memset(lw_net, 0, sizeof(*lw_net));
        }
    }
    const int NT = H / 16, NTD = H / 32;

    std::vector<int32_t> index;
    BlobLayout lay{};
    if (lw_net) {
        build_index_layerwise(index, d, lw_net);
    } else if (plan->generic_nt) {
        lay = build_index_generic(index, d, *plan);
    } else {
    std::vector<StepCols> enc_x, hid, hid_half, hid_skip_enc, dir_steps;
    encoding_steps(enc_x, FX, d.include_input_xyz != 0, 0);
    hidden_steps(hid, H, 0);
    hidden_steps(hid_half, H / 2, 0);
    encoding_steps(hid_skip_enc, FX, d.include_input_xyz != 0, H);  // cat(hidden, xyz): models.py:65
    hidden_steps(dir_steps, H, 0);
    encoding_steps(dir_steps, FD, d.include_input_dir != 0, H);     // cat(feat, view): models.py:72

    // ---- the blob as an index map: forward stream | biases | fc_alpha | fc_rgb | backward stream
    uint32_t skip_mask = 0;
    pack_gemm(index, T_L1W, dx, H, NT, enc_x);
    for (int i = 0; i < L - 1; ++i) {
        const bool skip = is_skip(d, i);
        const int ld = H + (skip ? dx : 0);
        pack_gemm(index, T_XYZ0 + 2 * i, ld, H, NT, hid);
        if (skip) {
            pack_gemm(index, T_XYZ0 + 2 * i, ld, H, NT, hid_skip_enc);
            skip_mask |= 1u << i;
        }
    }
    if (!no_view) {
        pack_gemm(index, T_FEATW, H, H, NT, hid);
        pack_gemm(index, T_DIRW, H + dd, H / 2, NTD, dir_steps);
    }
    index.resize(index.size() + 1024, -1);  // DMA granularity padding (4 KiB)
    pad_to(index, 64);
    const size_t off_bias = index.size();
    pack_range(index, T_L1B, H);
    for (int i = 0; i < L - 1; ++i) pack_range(index, T_XYZ0 + 2 * i + 1, H);
    if (no_view) index.resize(index.size() + H + H / 2, -1);       // the kernels' bias layout is the same for both kinds
    else {
        pack_range(index, T_FEATB, H);
        pack_range(index, T_DIRB, H / 2);
    }
    pack_range(index, T_ALPHAB, 1);
    pack_range(index, T_RGBB, 3);
    pad_to(index, 64);
    // fc_alpha / fc_rgb as per-lane-group GEMV operands
    const size_t off_wa = index.size();
    for (int g = 0; g < 4; ++g)
        for (int s = 0; s < H / 4; ++s) index.push_back((T_ALPHAW << 24) | (16 * (s >> 2) + 4 * g + (s & 3)));
    pad_to(index, 64);
    const size_t off_wr = index.size();
    if (no_view) {     // rows 0..2 of fc_out over the trunk output: three GEMV operands in fc_alpha's layout
        for (int c = 0; c < 3; ++c)
            for (int g = 0; g < 4; ++g)
                for (int s = 0; s < H / 4; ++s) index.push_back((T_RGBW << 24) | (c * H + 16 * (s >> 2) + 4 * g + (s & 3)));
    } else {
        for (int c = 0; c < 3; ++c)
            for (int g = 0; g < 4; ++g)
                for (int s = 0; s < H / 8; ++s)
                    index.push_back((T_RGBW << 24) | (c * (H / 2) + 16 * (s >> 2) + 4 * g + (s & 3)));
    }
    pad_to(index, 64);
    // backward (delta propagation): the same layers transposed, in reverse order, hidden columns only
    const size_t off_bwd = index.size();
    if (!no_view) {
        pack_gemm(index, T_DIRW, H + dd, H, NT, hid_half, true);
        pack_gemm(index, T_FEATW, H, H, NT, hid, true);
    }
    for (int i = L - 2; i >= 0; --i) pack_gemm(index, T_XYZ0 + 2 * i, H + (is_skip(d, i) ? dx : 0), H, NT, hid, true);
    index.resize(index.size() + 1024, -1);
    pad_to(index, 64);
    lay = BlobLayout{off_bias, off_wa, off_wr, off_bwd, skip_mask, 0, 0};
    }

    // opt-in bf16x3 stream: the same stages as units of (k-block, tile)
    std::vector<int32_t> index_b3;
    if (precision == NM_PREC_BF16X3) {
        if (!has_b3_kernel(H, FX, FD)) {
            set_error("precision bf16x3 is instantiated for hidden_size 64 / 128 / 256 with 6 or 10 xyz / 4 direction frequencies only");
            return 3;
        }
        std::vector<SlotCols> bx, bh, bskip, bdir;
        encoding_blocks(bx, FX, d.include_input_xyz != 0, 0, 2);
        hidden_blocks(bh, H, 0);
        encoding_blocks(bskip, FX, d.include_input_xyz != 0, H, 2);
        hidden_blocks(bdir, H, 0);
        encoding_blocks(bdir, FD, d.include_input_dir != 0, H, 1);
        pack_gemm_b3(index_b3, T_L1W, dx, H, NT, bx);
        for (int i = 0; i < L - 1; ++i) {
            const bool skip = is_skip(d, i);
            const int ld = H + (skip ? dx : 0);
            pack_gemm_b3(index_b3, T_XYZ0 + 2 * i, ld, H, NT, bh);
            if (skip) pack_gemm_b3(index_b3, T_XYZ0 + 2 * i, ld, H, NT, bskip);
        }
        pack_gemm_b3(index_b3, T_FEATW, H, H, NT, bh);
        pack_gemm_b3(index_b3, T_DIRW, H + dd, H / 2, NTD, bdir);
        index_b3.resize(index_b3.size() + 16 * 512, -1);      // DMA granularity / chunk padding
    }

    nm_mlp* m = new nm_mlp();
    std::memset(m, 0, sizeof(*m));
    m->desc = d;
    m->device = device;
    m->plan = plan;
    m->precision = precision;
    m->lw = lw_net;          // freed with the handle (also by the rollback below)
    if (lw_net) {
        for (int f = 0; f < FX; ++f) lw_net->bands_x[f] = w->freq_xyz[f];
        for (int f = 0; f < FD && !no_view; ++f) lw_net->bands_d[f] = w->freq_dir[f];
    }
    int prev_device = -1;
    (void)hipGetDevice(&prev_device);
    // every early return below (a failed hip call) gives back the half-built handle and the staging buffer and leaves the
    // caller's device current
    struct Rollback {
        nm_mlp* m; float* d_flat; int prev, dev; bool keep;
        ~Rollback() {
            if (d_flat) (void)hipFree(d_flat);
            if (!keep) nm_mlp_destroy(m);
            if (prev >= 0 && prev != dev) (void)hipSetDevice(prev);
        }
    } rollback{m, nullptr, prev_device, device, false};
    NM_HIP_CHECK(hipSetDevice(device));
    hipDeviceProp_t prop;
    NM_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    m->num_cus = prop.multiProcessorCount;
    // behind everything the kernels stream: PLAIN copies of the three tensors nm_mlp_linear_layer1_finish multiplies with, filled by
    // the same gather -- layers_xyz[0].weight (H, H) row-major, layer1.weight TRANSPOSED (dx, H), layer1.bias (H)
    if (!index.empty() && desc->num_layers >= 2) {
        const int Hh = desc->hidden_size, dxx = 6 * desc->num_encoding_fn_xyz + (desc->include_input_xyz ? 3 : 0);
        while (index.size() % 64) index.push_back(-1);
        m->plain_off = index.size();
        for (int e = 0; e < Hh * Hh; ++e) index.push_back((T_XYZ0 << 24) | e);
        for (int j = 0; j < dxx; ++j)
            for (int i = 0; i < Hh; ++i) index.push_back((T_L1W << 24) | (i * dxx + j));
        for (int i = 0; i < Hh; ++i) index.push_back((T_L1B << 24) | i);
    }
    m->blob_floats = index.size();
    m->blob_bytes = index.size() * 4;
    NM_HIP_CHECK(hipMalloc(&m->d_blob, m->blob_bytes));
    NM_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&m->d_check), 16));
    NM_HIP_CHECK(hipMemset(m->d_check, 0, 16));
    NM_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&m->d_index), m->blob_bytes));
    NM_HIP_CHECK(hipMemcpy(m->d_index, index.data(), m->blob_bytes, hipMemcpyHostToDevice));
    if (precision == NM_PREC_BF16X3) {
        m->b3_units = index_b3.size() / 512;
        NM_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&m->d_index_b3), index_b3.size() * 4));
        NM_HIP_CHECK(hipMemcpy(m->d_index_b3, index_b3.data(), index_b3.size() * 4, hipMemcpyHostToDevice));
        NM_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&m->d_tmp_b3), index_b3.size() * 4));
        NM_HIP_CHECK(hipMalloc(&m->d_stream_b3, m->b3_units * 3072));
    }
    const float* base = static_cast<const float*>(m->d_blob);
    MlpArgs& a = m->base;
    a.wstream = reinterpret_cast<const char*>(base);
    a.bias = base + lay.off_bias;
    a.walpha = base + lay.off_wa;
    a.wrgb = base + lay.off_wr;
    for (int f = 0; f < FX && f < MAX_FREQ_XYZ; ++f) a.bands_xyz[f] = w->freq_xyz[f];
    for (int f = 0; f < FD && f < MAX_FREQ_DIR && !no_view; ++f) a.bands_dir[f] = w->freq_dir[f];
    a.skip_mask = lay.skip_mask;
    if (plan->generic_nt) {      // the encodings' run-time description (mlp_device_g.h)
        const int parts = plan->variant == G_LONG_VARIANT ? G_ENC_PARTS : 1;     // [xyz, dir][parts][G_ENC_ARGS] (band, coordinate)
        float tab[2 * 2 * G_ENC_PARTS * G_ENC_ARGS];
        fill_enc_table(tab, parts, FX, w->freq_xyz);
        fill_enc_table(tab + 2 * parts * G_ENC_ARGS, parts, no_view ? 0 : FD, w->freq_dir);
        const size_t tab_bytes = sizeof(float) * 2 * 2 * parts * G_ENC_ARGS;
        NM_HIP_CHECK(hipMalloc(&m->d_enc_tab, tab_bytes));
        NM_HIP_CHECK(hipMemcpy(m->d_enc_tab, tab, tab_bytes, hipMemcpyHostToDevice));
        a.g_tab = m->d_enc_tab;
        a.g_nsx = (3 * FX + 1) / 2; a.g_idx = d.include_input_xyz ? 1 : 0; a.g_chx = lay.chx;
        a.g_nsd = no_view ? 0 : (3 * FD + 1) / 2; a.g_idd = (!no_view && d.include_input_dir) ? 1 : 0; a.g_chd = lay.chd;
        a.g_h = H; a.g_hd = H / 2;
        m->bwd.g_h = H; m->bwd.g_hd = H / 2;
    }
    m->bwd.wstream = reinterpret_cast<const char*>(base + lay.off_bwd);
    m->bwd.walpha = a.walpha;
    m->bwd.wrgb = a.wrgb;
    m->flops_full = 2 * mlp_macs(d, false);
    m->flops_density = 2 * mlp_macs(d, true);

    // ---- fill it from the host tensors: stage them on the device and run the same gather nm_mlp_refresh uses
    std::vector<float> flat;
    std::vector<size_t> offs(T_COUNT, 0);
    auto stage = [&](int t, const float* src, size_t count) { offs[t] = flat.size(); flat.insert(flat.end(), src, src + count); };
    stage(T_L1W, w->layer1_w, (size_t)H * dx); stage(T_L1B, w->layer1_b, H);
    for (int i = 0; i < L - 1; ++i) {
        stage(T_XYZ0 + 2 * i, w->layers_xyz_w[i], (size_t)H * (H + (is_skip(d, i) ? dx : 0)));
        stage(T_XYZ0 + 2 * i + 1, w->layers_xyz_b[i], H);
    }
    stage(T_ALPHAW, w->fc_alpha_w, H); stage(T_ALPHAB, w->fc_alpha_b, 1);
    if (no_view) stage(T_RGBW, w->fc_rgb_w, (size_t)3 * H);
    else {
        stage(T_FEATW, w->fc_feat_w, (size_t)H * H); stage(T_FEATB, w->fc_feat_b, H);
        stage(T_DIRW, w->layers_dir0_w, (size_t)(H / 2) * (H + dd)); stage(T_DIRB, w->layers_dir0_b, H / 2);
        stage(T_RGBW, w->fc_rgb_w, (size_t)3 * (H / 2));
    }
    stage(T_RGBB, w->fc_rgb_b, 3);
    NM_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&rollback.d_flat), flat.size() * 4));
    float* const d_flat = rollback.d_flat;
    NM_HIP_CHECK(hipMemcpy(d_flat, flat.data(), flat.size() * 4, hipMemcpyHostToDevice));
    WeightPtrs ptrs;
    std::memset(&ptrs, 0, sizeof(ptrs));
    ptrs.p[T_L1W] = d_flat + offs[T_L1W]; ptrs.p[T_L1B] = d_flat + offs[T_L1B];
    for (int t = T_XYZ0; t < T_XYZ0 + 2 * (L - 1); ++t) ptrs.p[t] = d_flat + offs[t];
    for (int t = T_FEATW; t < T_COUNT; ++t) ptrs.p[t] = d_flat + offs[t];
    int rc = launch_gather(m, ptrs, nullptr);
    if (rc == 0 && hipStreamSynchronize(nullptr) != hipSuccess) { set_error("parameter gather failed"); rc = 1; }
    if (rc) return rc;
    rollback.keep = true;
    *out = m;
    return 0;
}

int nm_mlp_refresh(nm_mlp* m, const nm_mlp_weights* d_weights, void* stream) {
    NM_REQUIRE(m && d_weights, "null argument");
    WeightPtrs ptrs;
    weight_pointers(m->desc, *d_weights, ptrs);
    for (int t = 0; t < T_COUNT; ++t) {
        const bool view_branch = t == T_FEATW || t == T_FEATB || t == T_DIRW || t == T_DIRB;
        const bool used = (t < T_XYZ0 + 2 * (m->desc.num_layers - 1) || t >= T_FEATW) && !(view_branch && !m->desc.use_viewdirs);
        NM_REQUIRE(!used || ptrs.p[t], "nm_mlp_refresh: missing tensor");
    }
    return launch_gather(m, ptrs, static_cast<hipStream_t>(stream));
}

int64_t nm_mlp_refresh_count(const nm_mlp* m) { return m ? m->refresh_count : -1; }

// Both products of the linearity identities in one launch: with S = [sums (H, dx) | colsum (H)]
//     l1w[i][j] = sum_k W0[k][i] sums[k][j],   l1b[i] = sum_k W0[k][i] colsum[k],   x0w[o][i] = sum_j sums[o][j] W1[i][j] + colsum[o] b1[i]
// one thread per output element, fixed summation order; W0, W1^T, b1 are the plain copies behind the packed image (coalesced or
// broadcast reads, everything L2-resident).  8.4 M multiply-adds at 8x256: launch-sized work -- it replaces two exports, three
// copies and two weight-gradient launches with their reductions (about 90 us of launches in an eager iteration).
__global__ __launch_bounds__(256) void linear_layer1_finish_kernel(const float* __restrict__ w0, const float* __restrict__ w1t,
                                                                   const float* __restrict__ b1, const float* __restrict__ sums, int ld,
                                                                   const float* __restrict__ colsum, int H, int dx, float* __restrict__ l1w,
                                                                   float* __restrict__ l1b, float* __restrict__ x0w) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int n_x0 = H * H;
    if (t < n_x0) {                       // x0w: thread (o, i), i fastest: sums[o][j] is a broadcast, w1t[j][i] coalesced
        const int o = t / H, i = t - o * H;
        float s = 0.0f;
#pragma unroll 8
        for (int j = 0; j < dx; ++j) s = fmaf(sums[(int64_t)o * ld + j], w1t[j * H + i], s);
        x0w[t] = fmaf(colsum[o], b1[i], s);
    } else if (t < n_x0 + (dx + 1) * H) { // l1: thread (j, i), i fastest: w0[k][i] coalesced, sums[k][j] a broadcast
        const int e = t - n_x0, j = e / H, i = e - j * H;
        float s = 0.0f;
        if (j < dx) {
#pragma unroll 8
            for (int k = 0; k < H; ++k) s = fmaf(w0[k * H + i], sums[(int64_t)k * ld + j], s);
            l1w[i * dx + j] = s;
        } else {
#pragma unroll 8
            for (int k = 0; k < H; ++k) s = fmaf(w0[k * H + i], colsum[k], s);
            l1b[i] = s;
        }
    }
}

int nm_mlp_linear_layer1_finish(nm_mlp* m, const float* d_sums, int32_t ld, const float* d_colsum, float* d_l1w, float* d_l1b,
                                float* d_x0w, void* stream_) {
    NM_REQUIRE(m && d_sums && d_colsum && d_l1w && d_l1b && d_x0w, "null argument");
    NM_REQUIRE(m->plain_off, "this handle keeps no plain copy of layer1 / layers_xyz[0] (layer-wise path, or a one-layer network)");
    const int H = m->desc.hidden_size, dx = 6 * m->desc.num_encoding_fn_xyz + (m->desc.include_input_xyz ? 3 : 0);
    NM_REQUIRE(ld >= dx, "row stride of the sums is smaller than the encoding");
    const float* w0 = static_cast<const float*>(m->d_blob) + m->plain_off;
    const int threads = H * H + (dx + 1) * H;
    hipLaunchKernelGGL(linear_layer1_finish_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream_),
                       w0, w0 + (size_t)H * H, w0 + (size_t)H * H + (size_t)dx * H, d_sums, (int)ld, d_colsum, H, dx, d_l1w, d_l1b, d_x0w);
    NM_HIP_CHECK(hipGetLastError());
    return 0;
}

int nm_mlp_export_layer1_transposed(nm_mlp* m, float* d_out, void* stream_) {
    NM_REQUIRE(m && d_out, "null argument");
    NM_REQUIRE(!m->lw, "the layer-wise path keeps no index map of its packed image");
    const int dx = 6 * m->desc.num_encoding_fn_xyz + (m->desc.include_input_xyz ? 3 : 0);
    const int64_t n = (int64_t)m->blob_floats;
    hipLaunchKernelGGL(scatter_layer1_t, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream_), m->d_index,
                       static_cast<const float*>(m->d_blob), n, (int)m->desc.hidden_size, dx, d_out);
    NM_HIP_CHECK(hipGetLastError());
    return 0;
}

int nm_mlp_export_xyz_weight(nm_mlp* m, int32_t layer, float* d_out, void* stream_) {
    NM_REQUIRE(m && d_out, "null argument");
    NM_REQUIRE(!m->lw, "the layer-wise path keeps no index map of its packed image");
    NM_REQUIRE(layer >= 0 && layer <= m->desc.num_layers - 2, "no such layers_xyz");
    const int64_t n = (int64_t)m->blob_floats;        // (launched on the caller's stream, which belongs to the handle's device)
    hipLaunchKernelGGL(scatter_tensor, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream_), m->d_index,
                       static_cast<const float*>(m->d_blob), n, (int)(T_XYZ0 + 2 * layer), d_out);
    NM_HIP_CHECK(hipGetLastError());
    return 0;
}

int nm_mlp_weights_current(nm_mlp* m, const nm_mlp_weights* d_weights, void* stream_, int32_t* differs) {
    NM_REQUIRE(m && d_weights && differs, "null argument");
    WeightPtrs ptrs;
    weight_pointers(m->desc, *d_weights, ptrs);
    for (int t = 0; t < T_COUNT; ++t) {
        const bool view_branch = t == T_FEATW || t == T_FEATB || t == T_DIRW || t == T_DIRB;
        const bool used = (t < T_XYZ0 + 2 * (m->desc.num_layers - 1) || t >= T_FEATW) && !(view_branch && !m->desc.use_viewdirs);
        NM_REQUIRE(!used || ptrs.p[t], "nm_mlp_weights_current: missing tensor");
    }
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int64_t n = (int64_t)m->blob_floats;
    NM_HIP_CHECK(hipMemsetAsync(m->d_check + 1, 0, 8, stream));
    hipLaunchKernelGGL(gather_parameters<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, m->d_index, ptrs,
                       static_cast<float*>(nullptr), n, m->d_check + 1);
    unsigned long long both[2];
    NM_HIP_CHECK(hipMemcpyAsync(both, m->d_check, 16, hipMemcpyDeviceToHost, stream));
    NM_HIP_CHECK(hipStreamSynchronize(stream));
    *differs = both[0] != both[1];
    return 0;
}

void nm_mlp_destroy(nm_mlp* m) {
    if (!m) return;
    if (m->d_blob) (void)hipFree(m->d_blob);
    if (m->d_index) (void)hipFree(m->d_index);
    if (m->d_check) (void)hipFree(m->d_check);
    if (m->d_index_b3) (void)hipFree(m->d_index_b3);
    if (m->d_tmp_b3) (void)hipFree(m->d_tmp_b3);
    if (m->d_stream_b3) (void)hipFree(m->d_stream_b3);
    if (m->d_enc_tab) (void)hipFree(m->d_enc_tab);
    layerwise_destroy(m);
    delete m;
}

int nm_mlp_kernel_variant(const nm_mlp* m, int* waves_per_workgroup) {
    int nw = 0;
    const int v = mlp_plan_info(m->plan, &nw);
    if (waves_per_workgroup) *waves_per_workgroup = nw;
    return v;
}

int64_t nm_mlp_flops_per_sample(const nm_mlp* m, int density_only) {
    return density_only ? m->flops_density : m->flops_full;
}

int nm_mlp_sample_points(nm_mlp* m, const float* d_points, const float* d_dirs, int64_t n, float* d_radiance,
                         void* stream) {
    NM_REQUIRE(m && d_points && d_dirs && d_radiance && n >= 0, "bad argument");
    MlpArgs a = m->base;
    a.mode = MODE_POINTS;
    a.a = d_points; a.b = d_dirs; a.c = nullptr;
    a.n = n; a.out = d_radiance;
    return launch_mlp_timed(m, a, 0, static_cast<hipStream_t>(stream));
}

int nm_mlp_eval_rays(nm_mlp* m, const float* d_origins, int origins_per_ray, const float* d_dirs, const float* d_t,
                     int64_t rays, int32_t samples, float* d_radiance, void* stream) {
    NM_REQUIRE(m && d_origins && d_dirs && d_t && d_radiance && rays >= 0 && samples > 0, "bad argument");
    MlpArgs a = m->base;
    a.mode = MODE_RAYS;
    a.a = d_origins; a.b = d_dirs; a.c = d_t;
    a.origins_per_ray = origins_per_ray; a.samples = samples;
    a.n = rays * samples; a.out = d_radiance;
    return launch_mlp_timed(m, a, 0, static_cast<hipStream_t>(stream));
}

}  // extern "C"

namespace nm {
int nm_mlp_eval_view_internal(nm_mlp* m, const RayGen* gen, const float* d_t, int64_t rays, int32_t samples,
                              float* d_radiance, hipStream_t stream) {
    NM_REQUIRE(m && gen && d_t && d_radiance && rays >= 0 && samples > 0, "bad argument");
    MlpArgs a = m->base;
    a.mode = MODE_VIEW;
    a.a = nullptr; a.b = nullptr; a.c = d_t;
    a.samples = samples; a.gen = *gen;
    a.n = rays * samples; a.out = d_radiance;
    return launch_mlp_timed(m, a, 0, stream);
}
}  // namespace nm

extern "C" {

int nm_mlp_grid_query(nm_mlp* m, const float* d_ax0, const float* d_ax1, const float* d_ax2, int32_t n0, int32_t n1,
                      int32_t n2, int64_t first, int64_t count, int32_t density_only, float* d_out, void* stream) {
    NM_REQUIRE(m && d_ax0 && d_ax1 && d_ax2 && d_out, "bad argument");
    NM_REQUIRE(n0 > 0 && n1 > 0 && n2 > 0 && first >= 0 && count >= 0 &&
               first + count <= (int64_t)n0 * n1 * n2, "grid range");
    MlpArgs a = m->base;
    a.mode = MODE_GRID;
    a.a = d_ax0; a.b = d_ax1; a.c = d_ax2;
    a.n1 = n1; a.n2 = n2; a.first = first;
    a.n = count; a.out = d_out;
    return launch_mlp_timed(m, a, density_only ? 1 : 0, static_cast<hipStream_t>(stream));
}

}  // extern "C"
