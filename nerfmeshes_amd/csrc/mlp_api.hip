// C ABI for the fused MLP: weight packing (torch.nn.Linear layout -> MFMA A-operand stream),
// handle lifetime, and the three entry points that launch nerf_mlp.hip's kernel.
#include <array>
#include <cstring>
#include <vector>

#include "nm_internal.h"

namespace nm {

static thread_local std::string g_error;
void set_error(const std::string& msg) { g_error = msg; }

const MlpPlan* find_mlp_plan(int H, int FX, int FD);
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

// Append the A-operand stream of one GEMM: for k-step s, block b of VW tiles, lane l, slot q:
// W[16*(VW*b+q) + (l&15)][cols[s][l>>4]].
static void pack_gemm(std::vector<float>& out, const float* W, int ld, int rows, int ntiles,
                      const std::vector<StepCols>& steps) {
    const int vw = ntiles >= 4 ? 4 : ntiles;
    for (const StepCols& c : steps)
        for (int b = 0; b < ntiles / vw; ++b)
            for (int l = 0; l < 64; ++l)
                for (int q = 0; q < vw; ++q) {
                    const int n = 16 * (vw * b + q) + (l & 15);
                    const int k = c[l >> 4];
                    out.push_back((n < rows && k >= 0) ? W[(size_t)n * ld + k] : 0.0f);
                }
}

// ---- optional per-launch timing of the dominant kernel (bench.py's roofline leg) ---------------
struct ProfRec { hipEvent_t start, stop; double flops; };
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof;

static int launch_mlp_timed(const nm_mlp* m, const MlpArgs& a, int density_only, hipStream_t stream) {
    if (!g_prof_on || a.n <= 0) return launch_mlp(m, a, density_only, stream);
    ProfRec r;
    NM_HIP_CHECK(hipEventCreate(&r.start));
    NM_HIP_CHECK(hipEventCreate(&r.stop));
    r.flops = (double)a.n * (double)(density_only ? m->flops_density : m->flops_full);
    NM_HIP_CHECK(hipEventRecord(r.start, stream));
    const int rc = launch_mlp(m, a, density_only, stream);
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
    macs += H;  // fc_alpha
    if (!density_only) macs += H * H + (H + dd) * (H / 2) + (H / 2) * 3;
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
    NM_REQUIRE(desc && w && out, "null argument");
    const nm_mlp_desc& d = *desc;
    NM_REQUIRE(d.use_viewdirs == 1, "only use_viewdirs=True networks are implemented on the HIP path");
    NM_REQUIRE(d.num_layers >= 2 && d.num_layers <= 32, "num_layers out of range");
    NM_REQUIRE(d.skip_step >= 1, "skip_step must be >= 1");
    NM_REQUIRE(d.num_encoding_fn_xyz <= MAX_FREQ_XYZ && d.num_encoding_fn_dir <= MAX_FREQ_DIR, "too many encoding fns");
    const MlpPlan* plan = find_mlp_plan(d.hidden_size, d.num_encoding_fn_xyz, d.num_encoding_fn_dir);
    if (!plan) {
        set_error("no gfx950 kernel instantiated for hidden_size=" + std::to_string(d.hidden_size) +
                  " num_encoding_fn_xyz=" + std::to_string(d.num_encoding_fn_xyz) +
                  " num_encoding_fn_dir=" + std::to_string(d.num_encoding_fn_dir) +
                  " (add a make_plan<> line in nerf_mlp.hip)");
        return 3;
    }
    const int H = d.hidden_size, L = d.num_layers, FX = d.num_encoding_fn_xyz, FD = d.num_encoding_fn_dir;
    const int dx = 6 * FX + (d.include_input_xyz ? 3 : 0), dd = 6 * FD + (d.include_input_dir ? 3 : 0);
    const int NT = H / 16, NTD = H / 32;

    std::vector<StepCols> enc_x, enc_d, hid, hid_skip_enc, dir_steps;
    encoding_steps(enc_x, FX, d.include_input_xyz != 0, 0);
    hidden_steps(hid, H, 0);
    encoding_steps(hid_skip_enc, FX, d.include_input_xyz != 0, H);  // cat(hidden, xyz): models.py:65
    hidden_steps(dir_steps, H, 0);
    encoding_steps(dir_steps, FD, d.include_input_dir != 0, H);     // cat(feat, view): models.py:72

    std::vector<float> stream;
    std::vector<float> bias;
    uint32_t skip_mask = 0;
    pack_gemm(stream, w->layer1_w, dx, H, NT, enc_x);
    bias.insert(bias.end(), w->layer1_b, w->layer1_b + H);
    for (int i = 0; i < L - 1; ++i) {
        const bool skip = i % d.skip_step == 0 && i > 0 && i != L - 1;  // models.py:37,63
        const int ld = H + (skip ? dx : 0);
        pack_gemm(stream, w->layers_xyz_w[i], ld, H, NT, hid);
        if (skip) {
            pack_gemm(stream, w->layers_xyz_w[i], ld, H, NT, hid_skip_enc);
            skip_mask |= 1u << i;
        }
        bias.insert(bias.end(), w->layers_xyz_b[i], w->layers_xyz_b[i] + H);
    }
    pack_gemm(stream, w->fc_feat_w, H, H, NT, hid);
    bias.insert(bias.end(), w->fc_feat_b, w->fc_feat_b + H);
    pack_gemm(stream, w->layers_dir0_w, H + dd, H / 2, NTD, dir_steps);
    bias.insert(bias.end(), w->layers_dir0_b, w->layers_dir0_b + H / 2);
    stream.resize(stream.size() + 1024, 0.0f);  // DMA granularity padding (4 KiB)

    // fc_alpha / fc_rgb as per-lane-group GEMV operands
    std::vector<float> walpha(4 * (H / 4)), wrgb(3 * 4 * (H / 8));
    for (int g = 0; g < 4; ++g)
        for (int s = 0; s < H / 4; ++s) walpha[g * (H / 4) + s] = w->fc_alpha_w[16 * (s >> 2) + 4 * g + (s & 3)];
    for (int c = 0; c < 3; ++c)
        for (int g = 0; g < 4; ++g)
            for (int s = 0; s < H / 8; ++s)
                wrgb[(c * 4 + g) * (H / 8) + s] = w->fc_rgb_w[(size_t)c * (H / 2) + 16 * (s >> 2) + 4 * g + (s & 3)];

    nm_mlp* m = new nm_mlp();
    std::memset(m, 0, sizeof(*m));
    m->desc = d;
    m->device = device;
    m->plan = plan;
    NM_HIP_CHECK(hipSetDevice(device));
    hipDeviceProp_t prop;
    NM_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    m->num_cus = prop.multiProcessorCount;
    auto align = [](size_t x) { return (x + 255) & ~size_t(255); };
    const size_t off_bias = align(stream.size() * 4), off_wa = off_bias + align(bias.size() * 4);
    const size_t off_wr = off_wa + align(walpha.size() * 4);
    m->blob_bytes = off_wr + align(wrgb.size() * 4);
    NM_HIP_CHECK(hipMalloc(&m->d_blob, m->blob_bytes));
    char* base = static_cast<char*>(m->d_blob);
    NM_HIP_CHECK(hipMemcpy(base, stream.data(), stream.size() * 4, hipMemcpyHostToDevice));
    NM_HIP_CHECK(hipMemcpy(base + off_bias, bias.data(), bias.size() * 4, hipMemcpyHostToDevice));
    NM_HIP_CHECK(hipMemcpy(base + off_wa, walpha.data(), walpha.size() * 4, hipMemcpyHostToDevice));
    NM_HIP_CHECK(hipMemcpy(base + off_wr, wrgb.data(), wrgb.size() * 4, hipMemcpyHostToDevice));
    MlpArgs& a = m->base;
    a.wstream = base;
    a.bias = reinterpret_cast<const float*>(base + off_bias);
    a.walpha = reinterpret_cast<const float*>(base + off_wa);
    a.wrgb = reinterpret_cast<const float*>(base + off_wr);
    a.balpha = w->fc_alpha_b[0];
    for (int c = 0; c < 3; ++c) a.brgb[c] = w->fc_rgb_b[c];
    for (int f = 0; f < FX; ++f) a.bands_xyz[f] = w->freq_xyz[f];
    for (int f = 0; f < FD; ++f) a.bands_dir[f] = w->freq_dir[f];
    a.skip_mask = skip_mask;
    m->flops_full = 2 * mlp_macs(d, false);
    m->flops_density = 2 * mlp_macs(d, true);
    *out = m;
    return 0;
}

void nm_mlp_destroy(nm_mlp* m) {
    if (!m) return;
    if (m->d_blob) (void)hipFree(m->d_blob);
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
