// FlexibleNeRFModel beyond the fused kernel families' limits -- hidden_size > 512, or an encoding of more than 48 k-steps (32 functions + input)
// (/root/reference/src/nerf/models.py:5-58 takes ANY hidden_size / num_encoding_fn_*; nm_mlp_create refused those until
// round 5) -- evaluated and trained LAYER BY LAYER on the general MFMA GEMM of nerf_dw_g.hip.
//
// Why not fused: the fused kernels keep a 16-sample tile's activations AND accumulators of a layer in one wavefront's
// registers (2 * 4 * NT for NT = hidden / 16 tiles): 512 wide is the register file.  Why layer by layer is fine HERE: a
// layer of a network this wide is 2 K N FLOP against 4 (K + N) bytes of activations per sample = 256 FLOP/B at 1024 wide -- far
// above the machine's 20 FLOP/B -- so materialising the activations costs bandwidth the GEMM does not need (the fused
// dataflow exists because a 256-wide layer at 64 FLOP/B does).
//
// Layout: activations are PLANAR (feature-major): X[k][i] = feature k of sample i of a batch, leading dimension `ld`.  A layer
// y = act(W x + b) is then C = A^T B with A = W^T (K x N, a second copy of the parameter in the handle's blob, re-gathered by
// nm_mlp_refresh like everything else) and B = X (K x batch): exactly the product dw_kernel_g computes -- a contraction over
// ROWS of two row-major operands -- with a short contraction and a wide output (dwg_gemm: one sample part, output blocks of up
// to 256 x 256 over grid y / z).  cat(x, xyz) / cat(feat, view) are two such products into two partial planes that the epilogue
// adds (models.py:64-65, :72: hidden columns first).  The backward (delta) chain is the same call with A = W itself
// (N x K: the contraction runs over the layer's outputs).  The epilogue (bias, ReLU / sigmoid, ReLU' mask) is a separate
// bandwidth-bound pass over the (N x batch) plane.
//
// Training keeps the ABI: nm_mlp_forward_train writes the tape rows [sample][feature] (transposed out of the planes, batch by
// batch), nm_mlp_backward reads them back for the ReLU' masks and writes the delta rows; the weight gradients are then
// nm_weight_grad_ex on those rows, as for every other handle.
#include <algorithm>

#include "nm_internal.h"
#include "mlp_device.h"
#include "nerf_layerwise.h"

namespace nm {

// ---- kernels ---------------------------------------------------------------------------------------------------------
struct LwEncArgs {
    int32_t fx, fd, inc_x, inc_d;
    float bands_x[LW_MAX_FREQ], bands_d[LW_MAX_FREQ];
};

// sample `first + i` of the call (any input mode: fetch_sample of the fused kernels) -> its PositionalEncoding columns
// (modules.py:26-34: [x | sin(x_c f_k), coordinate-major | cos(...)]) as planar rows; one thread per sample, so a
// wavefront's store of one encoding row is 256 contiguous bytes
__global__ __launch_bounds__(256) void lw_encode_kernel(const MlpArgs args, const LwEncArgs e, int64_t first, int count,
                                                        float* __restrict__ ex, float* __restrict__ ed, int64_t ld) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const SamplePD s = fetch_sample(args, first + i);
    const float p[3] = {s.px, s.py, s.pz}, d[3] = {s.dx, s.dy, s.dz};
    auto rows = [&](const float (&x)[3], int F, int inc, const float* bands, float* dst) {
        const int base = inc ? 3 : 0;
        if (inc) { dst[i] = x[0]; dst[ld + i] = x[1]; dst[2 * ld + i] = x[2]; }
        for (int a = 0; a < 3 * F; ++a) {
            float sv, cv;
            sincosf(x[a / F] * bands[a % F], &sv, &cv);
            dst[(int64_t)(base + a) * ld + i] = sv;
            dst[(int64_t)(base + 3 * F + a) * ld + i] = cv;
        }
    };
    rows(p, e.fx, e.inc_x, e.bands_x, ex);
    if (ed) rows(d, e.fd, e.inc_d, e.bands_d, ed);
}

enum LwAct : int { LW_NONE = 0, LW_RELU = 1, LW_SIGMOID = 2 };

// out[o][i] = act(sum_p partial[p][o][i] + bias[o]) (* [mask[o][i] > 0]) for o < rows, i < count
__global__ __launch_bounds__(256) void lw_epilogue_kernel(const float* __restrict__ partial, int parts, int64_t part_stride,
                                                          int64_t in_pad, const float* __restrict__ bias, int act,
                                                          const float* __restrict__ mask, int64_t mask_ld,
                                                          float* __restrict__ out, int64_t ld, int rows, int count) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int o = blockIdx.y;
    if (i >= count || o >= rows) return;
    const float* p = partial + (int64_t)o * in_pad + i;
    float v = p[0];
    for (int k = 1; k < parts; ++k) v += p[k * part_stride];
    if (bias) v += bias[o];
    if (act == LW_RELU) v = fmaxf(v, 0.0f);
    else if (act == LW_SIGMOID) v = 1.0f / (1.0f + expf(-v));
    if (mask && !(mask[(int64_t)o * mask_ld + i] > 0.0f)) v = 0.0f;
    out[(int64_t)o * ld + i] = v;
}

// (rgb (3 x ld) | sigma (ld)) planes -> radiance rows (n, 4), or sigma alone (density only)
__global__ __launch_bounds__(256) void lw_radiance_kernel(const float* __restrict__ rgb, const float* __restrict__ sigma,
                                                          int64_t ld, int64_t first, int count, int density_only,
                                                          float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    if (density_only) { out[first + i] = sigma[i]; return; }
    const f32x4 v = {rgb[i], rgb[ld + i], rgb[2 * ld + i], sigma[i]};
    reinterpret_cast<f32x4*>(out)[first + i] = v;
}

// plane (width x ld) <-> rows (n x width) of the samples first .. first + count: 32 x 32 tiles through LDS
__global__ __launch_bounds__(256) void lw_plane_to_rows_kernel(const float* __restrict__ plane, int64_t ld, int width, int count,
                                                               float* __restrict__ rows, int64_t first) {
    __shared__ float tile[32][33];
    const int i0 = blockIdx.x * 32, k0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8)
        if (k0 + r < width && i0 + tx < count) tile[r][tx] = plane[(int64_t)(k0 + r) * ld + i0 + tx];
    __syncthreads();
    for (int r = ty; r < 32; r += 8)
        if (i0 + r < count && k0 + tx < width) rows[(first + i0 + r) * width + k0 + tx] = tile[tx][r];
}

__global__ __launch_bounds__(256) void lw_rows_to_plane_kernel(const float* __restrict__ rows, int64_t first, int width, int count,
                                                               float* __restrict__ plane, int64_t ld) {
    __shared__ float tile[32][33];
    const int i0 = blockIdx.x * 32, k0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8)
        if (i0 + r < count && k0 + tx < width) tile[r][tx] = rows[(first + i0 + r) * width + k0 + tx];
    __syncthreads();
    for (int r = ty; r < 32; r += 8)
        if (k0 + r < width && i0 + tx < count) plane[(int64_t)(k0 + r) * ld + i0 + tx] = tile[tx][r];
}

// d_last (n, 4) = (g_rgb * s (1 - s), g_sigma) from the forward's radiance and its upstream gradient, also as a (4 x ld) plane
__global__ __launch_bounds__(256) void lw_dlast_kernel(const float* __restrict__ radiance, const float* __restrict__ grad,
                                                       int64_t first, int count, float* __restrict__ plane, int64_t ld,
                                                       float* __restrict__ d_last) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const f32x4 r = reinterpret_cast<const f32x4*>(radiance)[first + i], g = reinterpret_cast<const f32x4*>(grad)[first + i];
    f32x4 d;
#pragma unroll
    for (int c = 0; c < 3; ++c) d[c] = g[c] * (r[c] * (1.0f - r[c]));
    d[3] = g[3];
    reinterpret_cast<f32x4*>(d_last)[first + i] = d;
#pragma unroll
    for (int c = 0; c < 4; ++c) plane[c * ld + i] = d[c];
}

// ---- host side ---------------------------------------------------------------------------------------------------------
static int64_t round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

struct LwBuffers {
    float* ex; float* ed;              // encodings (dx / dd rows)
    std::vector<float*> h;             // activations: h[0] = layer1 output, h[1 + i] = relu(layers_xyz[i])
    float* feat; float* v;             // relu(fc_feat), relu(layers_dir[0])
    float* heads;                      // 4 rows: rgb[3] | sigma   (backward: the d_last plane)
    float* da; float* db;              // backward: two delta planes (ping-pong)
    float* partial;                    // 2 parts x out_pad_max x ld
    int64_t ld;
};

// the handle's workspace, grow-only; carved for a batch capacity `ld` (a multiple of 1024).  It lives IN the handle: a
// layer-wise handle is NOT re-entrant -- calls on it must be serialised by the caller (one stream / one thread at a time; the
// fused handles keep no per-call state and do not have this restriction).  Growing frees and re-allocates, which is illegal
// while `stream` is being captured into a hipGraph: the call then fails with a reason instead (warm the handle with the largest
// batch before the capture).
static int lw_carve(nm_mlp* m, LwNet* net, int64_t ld, bool training, LwBuffers* b, hipStream_t stream) {
    const int64_t H = net->H, H2 = net->flat ? 0 : net->H2, L = net->L;
    const int64_t out_pad_max = round_up(std::max<int64_t>(H, 16), 256);
    const int64_t planes_h = training ? L : 2;
    int64_t floats = (net->dx + net->dd + planes_h * H + (net->flat ? 0 : H + H2) + 4 + (training ? 2 * H : 0) + 2 * out_pad_max) * ld + 1024;
    if ((size_t)floats > net->ws_floats) {
        hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(stream, &capturing) == hipSuccess && capturing != hipStreamCaptureStatusNone) {
            set_error("layer-wise network path: the activation workspace would have to grow (" + std::to_string(floats * 4 >> 20) +
                      " MiB) while the stream is being captured: run this batch size once before the capture");
            return 1;
        }
        if (net->ws) (void)hipFree(net->ws);
        net->ws = nullptr; net->ws_floats = 0;
        void* p = nullptr;
        if (hipMalloc(&p, (size_t)floats * 4) != hipSuccess) {
            set_error("layer-wise network path: cannot allocate " + std::to_string(floats * 4 >> 20) + " MiB of activation workspace");
            return 1;
        }
        net->ws = static_cast<float*>(p); net->ws_floats = (size_t)floats;
    }
    float* c = net->ws;
    auto take = [&](int64_t rows) { float* p = c; c += rows * ld; return p; };
    b->ld = ld;
    b->ex = take(net->dx); b->ed = net->dd ? take(net->dd) : nullptr;
    b->h.clear();
    for (int l = 0; l < planes_h; ++l) b->h.push_back(take(H));
    b->feat = net->flat ? nullptr : take(H); b->v = net->flat ? nullptr : take(H2);
    b->heads = take(4);
    b->da = training ? take(H) : nullptr; b->db = training ? take(H) : nullptr;
    b->partial = take(2 * out_pad_max);
    (void)m;
    return 0;
}

// one layer: out (rows_out x count) = act(sum over the given (A, rows, B) products + bias), optionally masked
struct LwProduct { const float* A; int lda; int rows; const float* B; };
static int lw_layer(const LwBuffers& b, const LwProduct* prods, int nprods, int out, const float* bias, int act, const float* mask,
                    float* dst, int count, hipStream_t stream) {
    const DwgGemmGeometry g = dwg_gemm_geometry(out, count);
    const int64_t stride = (int64_t)g.out_pad * g.in_pad;
    int live = 0, only = -1;
    for (int k = 0; k < nprods; ++k)
        if (prods[k].rows > 0) { ++live; only = k; }
    if (live == 1) {       // a single product: bias / activation / mask straight from the GEMM's accumulators, no partial plane
        const DwgEpilogue epi{dst, bias, mask, b.ld, act};
        return dwg_gemm(prods[only].A, out, prods[only].lda, prods[only].B, count, b.ld, prods[only].rows, nullptr, stream, &epi);
    }
    int used = 0;
    for (int k = 0; k < nprods; ++k) {
        if (prods[k].rows <= 0) continue;
        if (int rc = dwg_gemm(prods[k].A, out, prods[k].lda, prods[k].B, count, b.ld, prods[k].rows, b.partial + used * stride, stream)) return rc;
        ++used;
    }
    NM_REQUIRE(used >= 1 && used <= 2, "layer-wise path: a layer is one or two products");
    hipLaunchKernelGGL(lw_epilogue_kernel, dim3((count + 255) / 256, out), dim3(256), 0, stream, b.partial, used, stride, g.in_pad, bias,
                       act, mask, b.ld, dst, b.ld, out, count);
    NM_HIP_CHECK(hipGetLastError());
    return 0;
}

static LwEncArgs lw_enc_args(const LwNet* net) {
    LwEncArgs e;
    e.fx = net->fx; e.fd = net->fd; e.inc_x = net->inc_x; e.inc_d = net->inc_d;
    for (int k = 0; k < LW_MAX_FREQ; ++k) { e.bands_x[k] = net->bands_x[k]; e.bands_d[k] = net->bands_d[k]; }
    return e;
}

// forward of one batch into the buffers; `keep`: every layer's activation in its own plane (training)
static int lw_forward_batch(const nm_mlp* m, const LwNet* net, const MlpArgs& args, int64_t first, int count, bool keep,
                            int density_only, const LwBuffers& b, hipStream_t stream) {
    const float* blob = static_cast<const float*>(m->d_blob);
    const int H = net->H, dx = net->dx, dd = net->dd, L = net->L;
    const bool want_dirs = !net->flat && density_only != 1 && dd > 0;
    hipLaunchKernelGGL(lw_encode_kernel, dim3((count + 255) / 256), dim3(256), 0, stream, args, lw_enc_args(net), first, count, b.ex,
                       want_dirs ? b.ed : nullptr, b.ld);
    NM_HIP_CHECK(hipGetLastError());
    auto plane = [&](int l) { return keep ? b.h[l] : b.h[l & 1]; };
    {   // layer1: no activation (models.py:62)
        const LwProduct p{blob + net->layer1.wt, H, dx, b.ex};
        if (int rc = lw_layer(b, &p, 1, H, blob + net->layer1.b, LW_NONE, nullptr, plane(0), count, stream)) return rc;
    }
    for (int i = 0; i < L - 1; ++i) {   // x = relu(layers_xyz[i](cat(x, xyz) at the skip layers)): models.py:63-66
        const LwLinear& lin = net->xyz[i];
        const bool skip = (net->skip_mask >> i) & 1u;
        const LwProduct p[2] = {{blob + lin.wt, H, H, plane(i)}, {blob + lin.wt + (size_t)H * H, H, skip ? dx : 0, b.ex}};
        if (int rc = lw_layer(b, p, 2, H, blob + lin.b, LW_RELU, nullptr, plane(i + 1), count, stream)) return rc;
    }
    const float* x = plane(L - 1);
    float* rgb = b.heads, *sigma = b.heads + 3 * b.ld;
    {   // fc_alpha on the trunk output (models.py:71; row 3 of fc_out for a network without view directions)
        const LwProduct p{blob + net->alpha.wt, 1, H, x};
        if (int rc = lw_layer(b, &p, 1, 1, blob + net->alpha.b, LW_NONE, nullptr, sigma, count, stream)) return rc;
    }
    if (density_only == 1) return 0;
    if (net->flat) {   // rows 0..2 of fc_out, sigmoid (models.py:77-79)
        const LwProduct p{blob + net->rgb.wt, 3, H, x};
        return lw_layer(b, &p, 1, 3, blob + net->rgb.b, LW_SIGMOID, nullptr, rgb, count, stream);
    }
    {   // feat = relu(fc_feat(x)) (models.py:70)
        const LwProduct p{blob + net->feat.wt, H, H, x};
        if (int rc = lw_layer(b, &p, 1, H, blob + net->feat.b, LW_RELU, nullptr, b.feat, count, stream)) return rc;
    }
    {   // v = relu(layers_dir[0](cat(feat, view))) (models.py:72-74)
        const LwProduct p[2] = {{blob + net->dir.wt, net->H2, H, b.feat}, {blob + net->dir.wt + (size_t)H * net->H2, net->H2, dd, b.ed}};
        if (int rc = lw_layer(b, p, 2, net->H2, blob + net->dir.b, LW_RELU, nullptr, b.v, count, stream)) return rc;
    }
    const LwProduct p{blob + net->rgb.wt, 3, net->H2, b.v};
    return lw_layer(b, &p, 1, 3, blob + net->rgb.b, LW_SIGMOID, nullptr, rgb, count, stream);
}

static int64_t lw_batch(const LwNet* net, int64_t n, bool training) {
    // a batch's planes must keep every operand below 2^32 bytes (dwg_gemm) and the workspace reasonable
    const int64_t kmax = std::max<int64_t>(net->H + std::max(net->dx, net->dd), 64);
    int64_t cap = std::min<int64_t>((0xe0000000ll / 4) / (kmax + 32), training ? 32768 : 65536);
    cap = std::max<int64_t>(cap / 1024 * 1024, 1024);
    return std::min(cap, round_up(n, 1024));
}

static void transpose_out(const float* plane, int64_t ld, int width, int count, float* rows, int64_t first, hipStream_t stream) {
    hipLaunchKernelGGL(lw_plane_to_rows_kernel, dim3((count + 31) / 32, (width + 31) / 32), dim3(256), 0, stream, plane, ld, width, count, rows, first);
}
static void transpose_in(const float* rows, int64_t first, int width, int count, float* plane, int64_t ld, hipStream_t stream) {
    hipLaunchKernelGGL(lw_rows_to_plane_kernel, dim3((count + 31) / 32, (width + 31) / 32), dim3(256), 0, stream, rows, first, width, count, plane, ld);
}

struct LwDeviceGuard {       // the handle's device current for the call (the stream belongs to it)
    int prev = -1;
    explicit LwDeviceGuard(int want) {
        int cur = -1;
        if (hipGetDevice(&cur) == hipSuccess && cur != want && hipSetDevice(want) == hipSuccess) prev = cur;
    }
    ~LwDeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

int layerwise_forward(nm_mlp* m, const MlpArgs& args, int density_only, hipStream_t stream) {
    LwNet* net = static_cast<LwNet*>(m->lw);
    NM_REQUIRE(net, "not a layer-wise handle");
    if (args.n <= 0) return 0;
    LwDeviceGuard guard(m->device);
    const int64_t ld = lw_batch(net, args.n, false);
    LwBuffers b;
    if (int rc = lw_carve(m, net, ld, false, &b, stream)) return rc;
    for (int64_t first = 0; first < args.n; first += ld) {
        const int count = (int)std::min<int64_t>(ld, args.n - first);
        if (int rc = lw_forward_batch(m, net, args, first, count, false, density_only == 1 ? 1 : 0, b, stream)) return rc;
        hipLaunchKernelGGL(lw_radiance_kernel, dim3((count + 255) / 256), dim3(256), 0, stream, b.heads, b.heads + 3 * b.ld, b.ld, first, count,
                           density_only == 1 ? 1 : 0, args.out);
        NM_HIP_CHECK(hipGetLastError());
    }
    return 0;
}

int layerwise_forward_train(nm_mlp* m, const MlpArgs& args, const nm_mlp_tape* tape, hipStream_t stream) {
    LwNet* net = static_cast<LwNet*>(m->lw);
    NM_REQUIRE(net, "not a layer-wise handle");
    // the weight-gradient planner tiles outputs of at most 8 blocks x 16 tiles x 16 = 2048 features per side (nerf_dw_g.hip):
    // say so where training starts, not three calls later as "no tiling"
    NM_REQUIRE(net->H <= 2048, "training: hidden_size > 2048 is inference-only (the weight-gradient kernels tile at most 2048 x 2048 outputs)");
    if (args.n <= 0) return 0;
    const int H = net->H, L = net->L;
    const int64_t ld = lw_batch(net, args.n, true);
    LwBuffers b;
    if (int rc = lw_carve(m, net, ld, true, &b, stream)) return rc;
    for (int64_t first = 0; first < args.n; first += ld) {
        const int count = (int)std::min<int64_t>(ld, args.n - first);
        if (int rc = lw_forward_batch(m, net, args, first, count, true, 0, b, stream)) return rc;
        hipLaunchKernelGGL(lw_radiance_kernel, dim3((count + 255) / 256), dim3(256), 0, stream, b.heads, b.heads + 3 * b.ld, b.ld, first, count, 0,
                           args.out);
        for (int l = 0; l < L; ++l) transpose_out(b.h[l], b.ld, H, count, tape->d_h + (int64_t)l * args.n * H, first, stream);
        if (!net->flat) {
            transpose_out(b.feat, b.ld, H, count, tape->d_feat, first, stream);
            transpose_out(b.v, b.ld, net->H2, count, tape->d_v, first, stream);
        }
        NM_HIP_CHECK(hipGetLastError());
    }
    return 0;
}

int layerwise_backward(nm_mlp* m, int64_t n, const nm_mlp_tape* tape, const float* d_radiance, const float* d_grad,
                       const nm_mlp_deltas* dl, hipStream_t stream) {
    LwNet* net = static_cast<LwNet*>(m->lw);
    NM_REQUIRE(net, "not a layer-wise handle");
    if (n <= 0) return 0;
    const float* blob = static_cast<const float*>(m->d_blob);
    const int H = net->H, H2 = net->H2, L = net->L, dx = net->dx, dd = net->dd;
    const int64_t ld = lw_batch(net, n, true);
    LwBuffers b;
    if (int rc = lw_carve(m, net, ld, true, &b, stream)) return rc;
    for (int64_t first = 0; first < n; first += ld) {
        const int count = (int)std::min<int64_t>(ld, n - first);
        // the ReLU' masks: the taped activations of this batch back as planes (layer1's output carries no ReLU)
        for (int l = 1; l < L; ++l) transpose_in(tape->d_h + (int64_t)l * n * H, first, H, count, b.h[l], b.ld, stream);
        if (!net->flat) {
            transpose_in(tape->d_feat, first, H, count, b.feat, b.ld, stream);
            transpose_in(tape->d_v, first, H2, count, b.v, b.ld, stream);
        }
        float* dlast = b.heads;     // rows 0..2: pre-sigmoid colour deltas, row 3: density delta
        hipLaunchKernelGGL(lw_dlast_kernel, dim3((count + 255) / 256), dim3(256), 0, stream, d_radiance, d_grad, first, count, dlast, b.ld,
                           dl->d_last);
        NM_HIP_CHECK(hipGetLastError());
        float* cur = b.da;          // delta at the pre-activation of the layer whose output is h[L - 1]
        float* nxt = b.db;
        if (net->flat) {            // fc_out^T applied to the four head deltas (models.py:77-79)
            const LwProduct p[2] = {{blob + net->rgb.w, H, 3, dlast}, {blob + net->alpha.w, H, 1, dlast + 3 * b.ld}};
            if (int rc = lw_layer(b, p, 2, H, nullptr, LW_NONE, L > 1 ? b.h[L - 1] : nullptr, cur, count, stream)) return rc;
        } else {
            {   // d v = fc_rgb^T d rgb, masked by v > 0
                const LwProduct p{blob + net->rgb.w, H2, 3, dlast};
                if (int rc = lw_layer(b, &p, 1, H2, nullptr, LW_NONE, b.v, nxt, count, stream)) return rc;
                transpose_out(nxt, b.ld, H2, count, dl->d_v, first, stream);
            }
            {   // d feat = layers_dir[0]^T[:, :H] d v, masked by feat > 0
                const LwProduct p{blob + net->dir.w, H + dd, H2, nxt};
                if (int rc = lw_layer(b, &p, 1, H, nullptr, LW_NONE, b.feat, cur, count, stream)) return rc;
                transpose_out(cur, b.ld, H, count, dl->d_feat, first, stream);
            }
            {   // d x = fc_feat^T d feat + fc_alpha^T d sigma, masked by the trunk output > 0
                const LwProduct p[2] = {{blob + net->feat.w, H, H, cur}, {blob + net->alpha.w, H, 1, dlast + 3 * b.ld}};
                if (int rc = lw_layer(b, p, 2, H, nullptr, LW_NONE, L > 1 ? b.h[L - 1] : nullptr, nxt, count, stream)) return rc;
                std::swap(cur, nxt);
            }
        }
        transpose_out(cur, b.ld, H, count, dl->d_h + (int64_t)(L - 1) * n * H, first, stream);
        for (int l = L - 2; l >= 0; --l) {      // through layers_xyz[l] (hidden columns only), masked by h[l] > 0 (l = 0: no ReLU)
            const LwLinear& lin = net->xyz[l];
            const bool skip = (net->skip_mask >> l) & 1u;
            const LwProduct p{blob + lin.w, H + (skip ? dx : 0), H, cur};
            if (int rc = lw_layer(b, &p, 1, H, nullptr, LW_NONE, l > 0 ? b.h[l] : nullptr, nxt, count, stream)) return rc;
            std::swap(cur, nxt);
            transpose_out(cur, b.ld, H, count, dl->d_h + (int64_t)l * n * H, first, stream);
        }
        NM_HIP_CHECK(hipGetLastError());
    }
    return 0;
}

void layerwise_destroy(nm_mlp* m) {
    LwNet* net = static_cast<LwNet*>(m->lw);
    if (!net) return;
    if (net->ws) (void)hipFree(net->ws);
    delete net;
    m->lw = nullptr;
}

}  // namespace nm
