// Per-ray primitives of the NeRF render path for gfx950: one 64-lane wavefront per ray, the
// sample axis (<= 256) lives in the wave, scans and reductions are wavefront shuffles.
//
// Replaces (reference file:line under /root/reference/src):
//   nerf/nerf_helpers.py:226-277  get_ray_bundle
//   nerf/modules.py:157-186       RaySampleInterval.forward (deterministic branch)
//   nerf/modules.py:67-121        VolumeRenderer.forward    (+ nerf_helpers.py:199-223 cumprod_exclusive)
//   nerf/modules.py:197-248       SamplePDF.forward / sample_pdf (deterministic u)
//
// All of these are HBM-bound streaming kernels (a few bytes in, a few bytes out per sample) and are
// < 1 % of the time of the path; they are written for exact agreement with the torch CPU
// semantics: torch.cumprod/cumsum accumulate in fp64 and round every output to fp32, torch.norm
// is an fma chain, a*b+c is two roundings (this file is compiled with -ffp-contract=off).
#include "nm_internal.h"

namespace nm {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

__device__ __forceinline__ double shfl_up_f64(double v, int delta) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_up(lo, delta);
    hi = __shfl_up(hi, delta);
    return __hiloint2double(hi, lo);
}

// exclusive multiplicative / additive scan across the 64 lanes, fp64
template <bool MUL>
__device__ __forceinline__ double wave_exclusive_scan(double v, int lane) {
    double incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const double o = shfl_up_f64(incl, off);
        if (lane >= off) incl = MUL ? incl * o : incl + o;
    }
    const double prev = shfl_up_f64(incl, 1);
    return lane == 0 ? (MUL ? 1.0 : 0.0) : prev;
}

__device__ __forceinline__ float torch_norm3(float x, float y, float z) { return nm_norm3(x, y, z); }

// ---- R0: pinhole rays -----------------------------------------------------------------------
struct Pose { float r[9]; };

__global__ void ray_bundle_kernel(Pose pose, int height, int width, float focal, int64_t first, int64_t count,
                                  float* __restrict__ dirs) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const int64_t pix = first + i;
    const int row = (int)(pix / width), colx = (int)(pix - (int64_t)row * width);
    const float x = ((float)colx - (float)(width * 0.5)) / focal;
    const float y = -((float)row - (float)(height * 0.5)) / focal;
    const float z = -1.0f;
    const float n = torch_norm3(x, y, z);
    const float dx = x / n, dy = y / n, dz = z / n;
#pragma unroll
    for (int a = 0; a < 3; ++a) dirs[3 * i + a] = (dx * pose.r[3 * a] + dy * pose.r[3 * a + 1]) + dz * pose.r[3 * a + 2];
}

__global__ void view_rays_kernel(const RayGen gen, int64_t rays, float* __restrict__ out_o, float* __restrict__ out_d) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rays) return;
    float o[3], d[3];
    nm_gen_ray(gen, i, o, d);
#pragma unroll
    for (int a = 0; a < 3; ++a) { out_o[3 * i + a] = o[a]; out_d[3 * i + a] = d[a]; }
}

// ndc_rays (nerf_helpers.py:280-307) over n rays; origins (1,3) shared or (n,3)
__global__ void ndc_rays_kernel(const float* __restrict__ origins, int origins_per_ray, const float* __restrict__ dirs,
                                int64_t n, float near32, float c_w, float c_h, float two_near, float m_two_near,
                                float* __restrict__ out_o, float* __restrict__ out_d) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* op = origins + (origins_per_ray ? 3 * i : 0);
    float o[3] = {op[0], op[1], op[2]}, d[3] = {dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]};
    nm_ndc_ray(near32, c_w, c_h, two_near, m_two_near, o, d);
#pragma unroll
    for (int a = 0; a < 3; ++a) { out_o[3 * i + a] = o[a]; out_d[3 * i + a] = d[a]; }
}

// PositionalEncoding.forward (modules.py:26-34) on (n, dim) rows: [x | sin(x_c * f_k) c-major | cos(...)]
struct BandArgs { float f[MAX_FREQ_XYZ]; };
__global__ void positional_encoding_kernel(const float* __restrict__ x, int64_t n, int dim, int nf, int include_input,
                                           BandArgs bands, float* __restrict__ out) {
    const int width = 2 * dim * nf + (include_input ? dim : 0);
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * (int64_t)(dim * nf)) return;
    const int64_t row = i / (dim * nf);
    const int a = (int)(i - row * (dim * nf));       // a = coord * nf + freq
    const int c = a / nf, k = a - c * nf;
    const float v = x[row * dim + c];
    float* o = out + row * width;
    const float arg = bands.f[k] * v;
    float sv, cv;
    sincosf(arg, &sv, &cv);
    const int base = include_input ? dim : 0;
    o[base + a] = sv;
    o[base + dim * nf + a] = cv;
    if (include_input && k == 0) o[c] = v;
}

// ---- R1: coarse depth samples -----------------------------------------------------------------
__global__ void coarse_intervals_kernel(const float* __restrict__ u, const float* __restrict__ near_,
                                        const float* __restrict__ far_, int per_ray, int lindisp, int64_t rays,
                                        int samples, float* __restrict__ t) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rays * samples) return;
    const int64_t ray = i / samples;
    const int k = (int)(i - ray * samples);
    const float nr = near_[per_ray ? ray : 0], fr = far_[per_ray ? ray : 0];
    const float uk = u[k], om = 1.0f - uk;
    float v;
    if (!lindisp) {
        v = nr * om + fr * uk;
    } else {
        const float a = (1.0f / nr) * om, b = (1.0f / fr) * uk;
        v = 1.0f / (a + b);
    }
    t[i] = v;
}

// ---- R4: alpha compositing ----------------------------------------------------------------------
// One wave per ray; lane owns PER consecutive samples.
template <int PER>
__global__ __launch_bounds__(256) void composite_kernel(const float* __restrict__ radiance,
                                                        const float* __restrict__ t, const float* __restrict__ dirs,
                                                        const float* __restrict__ noise, int64_t rays, int samples,
                                                        float thr, int white_bg, int training, nm_bundle_out out,
                                                        const RayGen gen) {
    const int lane = threadIdx.x & 63;
    const int64_t ray = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (ray >= rays) return;
    float norm;
    if (gen.enabled) {       // rays generated from the pose: no direction buffer exists
        float go[3], gd[3];
        nm_gen_ray(gen, ray, go, gd);
        norm = torch_norm3(gd[0], gd[1], gd[2]);
    } else {
        norm = torch_norm3(dirs[3 * ray], dirs[3 * ray + 1], dirs[3 * ray + 2]);
    }
    const float* tr = t + ray * samples;
    const f32x4* rr = reinterpret_cast<const f32x4*>(radiance) + ray * samples;

    float alpha[PER], tt[PER];
    f32x4 rad[PER];
    double local = 1.0;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const int s = lane * PER + q;
        alpha[q] = 0.0f; tt[q] = 0.0f; rad[q] = f32x4{0, 0, 0, 0};
        if (s < samples) {
            tt[q] = tr[s];
            rad[q] = rr[s];
            float dist = (s + 1 < samples) ? (tr[s + 1] - tt[q]) : 1e10f;
            dist = dist * norm;
            const float sig = fmaxf(rad[q][3] + (noise ? noise[ray * samples + s] : 0.0f), 0.0f);   // modules.py:82-93
            alpha[q] = 1.0f - expf(-sig * dist);
            local *= (double)((1.0f - alpha[q]) + 1e-10f);
        }
    }
    double run = wave_exclusive_scan<true>(local, lane);  // product of everything before this lane
    float r = 0, g = 0, b = 0, acc = 0, depth = 0;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const int s = lane * PER + q;
        if (s < samples) {
            const float T = (float)run;           // fp64 running product rounded per element, as torch.cumprod
            const float w = alpha[q] * T;
            if (out.d_weights) out.d_weights[ray * samples + s] = w;
            if (out.d_mask_weights) out.d_mask_weights[ray * samples + s] = T > thr ? 1.0f : 0.0f;
            r += w * rad[q][0]; g += w * rad[q][1]; b += w * rad[q][2];
            acc += w;
            depth += w * tt[q];
            run *= (double)((1.0f - alpha[q]) + 1e-10f);
        }
    }
    r = wave_sum(r); g = wave_sum(g); b = wave_sum(b); acc = wave_sum(acc); depth = wave_sum(depth);
    if (lane == 0) {
        float disp = 1.0f / fmaxf(1e-10f, depth / acc);
        if (disp != disp || (depth / acc) != (depth / acc)) disp = 0.0f;  // NaN -> 0 (modules.py:107)
        if (!training && acc < 1.0f) depth = 0.0f;                        // modules.py:108-109
        if (white_bg) { const float bg = 1.0f - acc; r += bg; g += bg; b += bg; }
        if (out.d_rgb_map) { out.d_rgb_map[3 * ray] = r; out.d_rgb_map[3 * ray + 1] = g; out.d_rgb_map[3 * ray + 2] = b; }
        if (out.d_depth_map) out.d_depth_map[ray] = depth;
        if (out.d_acc_map) out.d_acc_map[ray] = acc;
        if (out.d_disp_map) out.d_disp_map[ray] = disp;
    }
}

// ---- R5: inverse-CDF resampling + merge -------------------------------------------------------------
// One wave per ray, its cdf / bins / merged depths in LDS: 3 * coarse + fine floats per wave, sized per launch (192-sample
// rays: 1.3 KB per wave; the limit is what four waves fit into a CU's 160 KB: 3 * num_coarse + num_fine <= 10 240).
constexpr int PDF_MAX_FLOATS_PER_WAVE = 10240;

__global__ __launch_bounds__(256) void sample_pdf_kernel(const float* __restrict__ t, const float* __restrict__ weights,
                                                         const float* __restrict__ u, int u_per_ray, int64_t rays,
                                                         int coarse, int fine, float* __restrict__ t_out) {
    extern __shared__ __attribute__((aligned(16))) float s_pdf[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t ray = (int64_t)blockIdx.x * 4 + wv;
    if (ray >= rays) return;  // wave-uniform; no block-level barriers below
    float* cdf = s_pdf + (size_t)wv * (3 * coarse + fine);
    float* bins = cdf + coarse;
    float* all = bins + coarse;
    const float* tr = t + ray * coarse;
    const float* wr = weights + ray * coarse;
    const int nb = coarse - 1;   // bins / cdf entries
    const int np = coarse - 2;   // pdf entries (weights[1:-1])

    // bins, pdf numerators, their sum
    float psum = 0.0f;
    for (int i = lane; i < coarse; i += 64) {
        const float ti = tr[i];
        all[i] = ti;
        if (i < nb) bins[i] = 0.5f * (tr[i + 1] + ti);
        if (i < np) psum += wr[i + 1] + 1e-5f;
    }
    psum = wave_sum(psum);
    // cdf = [0, cumsum(pdf)]: fp64 accumulation rounded per element (torch.cumsum on CPU)
    double carry = 0.0;
    for (int base = 0; base < np; base += 64) {
        const int i = base + lane;
        const double pdf = i < np ? (double)((wr[i + 1] + 1e-5f) / psum) : 0.0;
        double incl = pdf;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const double o = shfl_up_f64(incl, off);
            if (lane >= off) incl += o;
        }
        incl += carry;
        if (i < np) cdf[i + 1] = (float)incl;
        int lo = __double2loint(incl), hi = __double2hiint(incl);
        carry = __hiloint2double(__shfl(hi, 63), __shfl(lo, 63));
    }
    if (lane == 0) cdf[0] = 0.0f;
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();

    // invert: idx = searchsorted(cdf, u, right=True)
    for (int j = lane; j < fine; j += 64) {
        const float uj = u[(u_per_ray ? ray * fine : 0) + j];   // modules.py:221-228: linspace, or rand per ray
        int lo = 0, hi = nb;             // first index with cdf[idx] > uj
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (cdf[mid] <= uj) lo = mid + 1; else hi = mid;
        }
        const int below = lo - 1 > 0 ? lo - 1 : 0;
        const int above = lo < nb - 1 ? lo : nb - 1;
        const float cb = cdf[below], ca = cdf[above];
        float denom = ca - cb;
        if (denom < 1e-5f) denom = 1.0f;
        const float frac = (uj - cb) / denom;
        const float bb = bins[below], ba = bins[above];
        all[coarse + j] = bb + frac * (ba - bb);
    }
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();

    // sort(cat(t, samples)): rank sort (any order of inputs, ties by index) -- 192 elements per ray
    const int total = coarse + fine;
    float* dst = t_out + ray * total;
    for (int e = lane; e < total; e += 64) {
        const float v = all[e];
        int rank = 0;
        for (int k = 0; k < total; ++k) {
            const float o = all[k];
            rank += (o < v || (o == v && k < e)) ? 1 : 0;
        }
        dst[rank] = v;
    }
}

// ---- training-mode pieces -----------------------------------------------------------
// Stratified jitter of the coarse depths (modules.py:171-184): lower + (upper - lower) * rand, with
// mids = 0.5 * (t[k+1] + t[k]); the random numbers are the caller's (torch.rand on the device).
__global__ void perturb_intervals_kernel(const float* __restrict__ t, const float* __restrict__ rnd, int64_t rays,
                                         int samples, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rays * samples) return;
    const int k = (int)(i % samples);
    const float tk = t[i];
    const float upper = k + 1 < samples ? 0.5f * (t[i + 1] + tk) : tk;
    const float lower = k > 0 ? 0.5f * (tk + t[i - 1]) : tk;
    const float span = upper - lower;
    const float step = span * rnd[i];
    out[i] = lower + step;
}

// Backward of VolumeRenderer.forward (modules.py:67-121) for the differentiable outputs the losses use:
//   w_k = alpha_k * T_k,  T_k = prod_{j<k} (1 - alpha_j + 1e-10),  alpha_k = 1 - exp(-relu(sigma_k + noise_k) * dist_k)
//   dL/dw_k = G_rgb . rgb_k + G_acc + G_depth * t_k + G_w[k] (- sum(G_rgb) with a white background)
//   dL/dalpha_k = dL/dw_k * T_k - (sum_{j>k} dL/dw_j * w_j) / (1 - alpha_k + 1e-10)
// One wave per ray, lane owns PER consecutive samples (the forward kernel's layout): T_k from the same fp64 exclusive
// product scan as the forward, the suffix sums sum_{j>k} dL/dw_j * w_j from an fp64 suffix scan (reverse lane order).
// (Until round 5: one THREAD per ray, two sequential sweeps over its samples with T_k / alpha_k parked in the output rows --
// 2048 threads for a 2048-ray batch, 0.12 ms per call, 0.25 ms of every training iteration.)
template <int PER>
__global__ __launch_bounds__(256) void composite_backward_kernel(const float* __restrict__ radiance,
                                                                 const float* __restrict__ t,
                                                                 const float* __restrict__ dirs,
                                                                 const float* __restrict__ noise, int64_t rays, int samples,
                                                                 int white_bg, nm_bundle_grads g,
                                                                 float* __restrict__ grad_radiance) {
    const int lane = threadIdx.x & 63;
    const int64_t ray = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (ray >= rays) return;
    const float norm = torch_norm3(dirs[3 * ray], dirs[3 * ray + 1], dirs[3 * ray + 2]);
    const float* tr = t + ray * samples;
    const f32x4* rr = reinterpret_cast<const f32x4*>(radiance) + ray * samples;
    f32x4* out = reinterpret_cast<f32x4*>(grad_radiance) + ray * samples;
    float gr = 0.0f, gg = 0.0f, gb = 0.0f;
    if (g.d_rgb_map) { gr = g.d_rgb_map[3 * ray]; gg = g.d_rgb_map[3 * ray + 1]; gb = g.d_rgb_map[3 * ray + 2]; }
    const float gacc = (g.d_acc_map ? g.d_acc_map[ray] : 0.0f) - (white_bg ? (gr + gg + gb) : 0.0f);
    const float gdepth = g.d_depth_map ? g.d_depth_map[ray] : 0.0f;

    float alpha[PER], dist[PER], dw[PER], raw[PER];
    f32x4 rad[PER];
    double local = 1.0;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const int s = lane * PER + q;
        alpha[q] = 0.0f; dist[q] = 0.0f; dw[q] = 0.0f; raw[q] = 0.0f; rad[q] = f32x4{0, 0, 0, 0};
        if (s < samples) {
            const float ts = tr[s];
            rad[q] = rr[s];
            float d = (s + 1 < samples) ? (tr[s + 1] - ts) : 1e10f;
            d = d * norm;
            dist[q] = d;
            raw[q] = rad[q][3] + (noise ? noise[ray * samples + s] : 0.0f);
            alpha[q] = 1.0f - expf(-fmaxf(raw[q], 0.0f) * d);
            local *= (double)((1.0f - alpha[q]) + 1e-10f);
            dw[q] = (gr * rad[q][0] + gg * rad[q][1] + gb * rad[q][2]) + gacc + gdepth * ts;
            if (g.d_weights) dw[q] += g.d_weights[ray * samples + s];
        }
    }
    double run = wave_exclusive_scan<true>(local, lane);      // product of everything before this lane
    float T[PER];
    double own = 0.0;                                          // sum of dL/dw_j * w_j over this lane's samples
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        T[q] = (float)run;                                     // fp64 running product rounded per element, as the forward
        run *= (double)((1.0f - alpha[q]) + 1e-10f);
        own += (double)(dw[q] * (alpha[q] * T[q]));
    }
    // suffix over the lanes above this one: exclusive additive scan in reversed lane order
    double after = wave_exclusive_scan<false>(__shfl(own, 63 - lane), lane);
    double suffix = __shfl(after, 63 - lane);
#pragma unroll
    for (int q = PER - 1; q >= 0; --q) {
        const int s = lane * PER + q;
        if (s < samples) {
            const float w = alpha[q] * T[q];
            const float keep = (1.0f - alpha[q]) + 1e-10f;
            const float dalpha = dw[q] * T[q] - (float)suffix / keep;
            suffix += (double)(dw[q] * w);
            const float dsigma = raw[q] > 0.0f ? dalpha * dist[q] * (1.0f - alpha[q]) : 0.0f;
            out[s] = f32x4{w * gr, w * gg, w * gb, dsigma};
        }
    }
}

// ---- rays of MORE than 512 samples (no shipped config has them; /root/reference/src/nerf/modules.py:67-121 takes any count):
// the same arithmetic in segments of 512 samples (8 per lane), the transmittance carried from segment to segment in fp64.
// The <= 512 kernels above keep their one-pass form (and their bits).
__device__ __forceinline__ double shfl_f64(double v, int src) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    return __hiloint2double(__shfl(hi, src), __shfl(lo, src));
}

__global__ __launch_bounds__(256) void composite_long_kernel(const float* __restrict__ radiance, const float* __restrict__ t,
                                                             const float* __restrict__ dirs, const float* __restrict__ noise,
                                                             int64_t rays, int samples, float thr, int white_bg, int training,
                                                             nm_bundle_out out, const RayGen gen) {
    constexpr int PER = 8;
    const int lane = threadIdx.x & 63;
    const int64_t ray = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (ray >= rays) return;
    float norm;
    if (gen.enabled) {
        float go[3], gd[3];
        nm_gen_ray(gen, ray, go, gd);
        norm = torch_norm3(gd[0], gd[1], gd[2]);
    } else {
        norm = torch_norm3(dirs[3 * ray], dirs[3 * ray + 1], dirs[3 * ray + 2]);
    }
    const float* tr = t + ray * samples;
    const f32x4* rr = reinterpret_cast<const f32x4*>(radiance) + ray * samples;
    double carry = 1.0;                                       // product of (1 - alpha + 1e-10) over every earlier segment
    float r = 0, g = 0, b = 0, acc = 0, depth = 0;
    for (int base = 0; base < samples; base += 64 * PER) {
        float alpha[PER], tt[PER];
        f32x4 rad[PER];
        double local = 1.0;
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int s = base + lane * PER + q;
            alpha[q] = 0.0f; tt[q] = 0.0f; rad[q] = f32x4{0, 0, 0, 0};
            if (s < samples) {
                tt[q] = tr[s];
                rad[q] = rr[s];
                float dist = (s + 1 < samples) ? (tr[s + 1] - tt[q]) : 1e10f;
                dist = dist * norm;
                const float sig = fmaxf(rad[q][3] + (noise ? noise[ray * samples + s] : 0.0f), 0.0f);
                alpha[q] = 1.0f - expf(-sig * dist);
                local *= (double)((1.0f - alpha[q]) + 1e-10f);
            }
        }
        double run = carry * wave_exclusive_scan<true>(local, lane);
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int s = base + lane * PER + q;
            if (s < samples) {
                const float T = (float)run;
                const float w = alpha[q] * T;
                if (out.d_weights) out.d_weights[ray * samples + s] = w;
                if (out.d_mask_weights) out.d_mask_weights[ray * samples + s] = T > thr ? 1.0f : 0.0f;
                r += w * rad[q][0]; g += w * rad[q][1]; b += w * rad[q][2];
                acc += w;
                depth += w * tt[q];
                run *= (double)((1.0f - alpha[q]) + 1e-10f);
            }
        }
        carry = shfl_f64(run, 63);                            // lane 63 has walked to the end of the segment
    }
    r = wave_sum(r); g = wave_sum(g); b = wave_sum(b); acc = wave_sum(acc); depth = wave_sum(depth);
    if (lane == 0) {
        float disp = 1.0f / fmaxf(1e-10f, depth / acc);
        if (disp != disp || (depth / acc) != (depth / acc)) disp = 0.0f;
        if (!training && acc < 1.0f) depth = 0.0f;
        if (white_bg) { const float bg = 1.0f - acc; r += bg; g += bg; b += bg; }
        if (out.d_rgb_map) { out.d_rgb_map[3 * ray] = r; out.d_rgb_map[3 * ray + 1] = g; out.d_rgb_map[3 * ray + 2] = b; }
        if (out.d_depth_map) out.d_depth_map[ray] = depth;
        if (out.d_acc_map) out.d_acc_map[ray] = acc;
        if (out.d_disp_map) out.d_disp_map[ray] = disp;
    }
}

// backward, two sweeps: the first yields total = sum_j dL/dw_j * w_j, the second walks the samples again with the running
// prefix, so that the suffix sum_{j>k} of the one-pass kernel is total - prefix_k - dL/dw_k * w_k (all fp64)
__global__ __launch_bounds__(256) void composite_backward_long_kernel(const float* __restrict__ radiance, const float* __restrict__ t,
                                                                      const float* __restrict__ dirs, const float* __restrict__ noise,
                                                                      int64_t rays, int samples, int white_bg, nm_bundle_grads g,
                                                                      float* __restrict__ grad_radiance) {
    constexpr int PER = 8;
    const int lane = threadIdx.x & 63;
    const int64_t ray = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (ray >= rays) return;
    const float norm = torch_norm3(dirs[3 * ray], dirs[3 * ray + 1], dirs[3 * ray + 2]);
    const float* tr = t + ray * samples;
    const f32x4* rr = reinterpret_cast<const f32x4*>(radiance) + ray * samples;
    f32x4* out = reinterpret_cast<f32x4*>(grad_radiance) + ray * samples;
    float gr = 0.0f, gg = 0.0f, gb = 0.0f;
    if (g.d_rgb_map) { gr = g.d_rgb_map[3 * ray]; gg = g.d_rgb_map[3 * ray + 1]; gb = g.d_rgb_map[3 * ray + 2]; }
    const float gacc = (g.d_acc_map ? g.d_acc_map[ray] : 0.0f) - (white_bg ? (gr + gg + gb) : 0.0f);
    const float gdepth = g.d_depth_map ? g.d_depth_map[ray] : 0.0f;
    double total = 0.0;
    for (int sweep = 0; sweep < 2; ++sweep) {
        double carry = 1.0, prefix = 0.0;                     // transmittance / sum of dL/dw_j * w_j in front of this segment
        for (int base = 0; base < samples; base += 64 * PER) {
            float alpha[PER], dist[PER], dw[PER], raw[PER];
            f32x4 rad[PER];
            double local = 1.0;
#pragma unroll
            for (int q = 0; q < PER; ++q) {
                const int s = base + lane * PER + q;
                alpha[q] = 0.0f; dist[q] = 0.0f; dw[q] = 0.0f; raw[q] = 0.0f; rad[q] = f32x4{0, 0, 0, 0};
                if (s < samples) {
                    const float ts = tr[s];
                    rad[q] = rr[s];
                    float d = (s + 1 < samples) ? (tr[s + 1] - ts) : 1e10f;
                    d = d * norm;
                    dist[q] = d;
                    raw[q] = rad[q][3] + (noise ? noise[ray * samples + s] : 0.0f);
                    alpha[q] = 1.0f - expf(-fmaxf(raw[q], 0.0f) * d);
                    local *= (double)((1.0f - alpha[q]) + 1e-10f);
                    dw[q] = (gr * rad[q][0] + gg * rad[q][1] + gb * rad[q][2]) + gacc + gdepth * ts;
                    if (g.d_weights) dw[q] += g.d_weights[ray * samples + s];
                }
            }
            double run = carry * wave_exclusive_scan<true>(local, lane);
            float T[PER];
            double own = 0.0;
#pragma unroll
            for (int q = 0; q < PER; ++q) {
                T[q] = (float)run;
                run *= (double)((1.0f - alpha[q]) + 1e-10f);
                own += (double)(dw[q] * (alpha[q] * T[q]));
            }
            carry = shfl_f64(run, 63);
            const double before = prefix + wave_exclusive_scan<false>(own, lane);    // everything in front of this lane's samples
            prefix = shfl_f64(before + own, 63);
            if (sweep == 0) continue;
            double done = before;
#pragma unroll
            for (int q = 0; q < PER; ++q) {
                const int s = base + lane * PER + q;
                if (s < samples) {
                    const float w = alpha[q] * T[q];
                    const float keep = (1.0f - alpha[q]) + 1e-10f;
                    done += (double)(dw[q] * w);
                    const float dalpha = dw[q] * T[q] - (float)(total - done) / keep;
                    const float dsigma = raw[q] > 0.0f ? dalpha * dist[q] * (1.0f - alpha[q]) : 0.0f;
                    out[s] = f32x4{w * gr, w * gg, w * gb, dsigma};
                }
            }
        }
        total = prefix;
    }
}

// ---- launchers ------------------------------------------------------------------------------------
int launch_ray_bundle(const float* c2w, int height, int width, float focal, int64_t first, int64_t count, float* d_dirs,
                      hipStream_t stream) {
    if (count <= 0) return 0;
    Pose p;
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) p.r[3 * a + b] = c2w[4 * a + b];
    hipLaunchKernelGGL(ray_bundle_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, stream, p, height, width,
                       focal, first, count, d_dirs);
    NM_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_coarse_intervals(const float* d_u, const float* d_near, const float* d_far, int per_ray, int lindisp,
                            int64_t rays, int samples, float* d_t, hipStream_t stream) {
    const int64_t n = rays * samples;
    if (n <= 0) return 0;
    hipLaunchKernelGGL(coarse_intervals_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, d_u, d_near,
                       d_far, per_ray, lindisp, rays, samples, d_t);
    NM_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_composite(const float* d_radiance, const float* d_t, const float* d_dirs, const float* d_noise, int64_t rays,
                     int samples, float thr, int white_bg, int training, const nm_bundle_out& out, hipStream_t stream,
                     const RayGen* gen_ptr = nullptr) {
    if (rays <= 0) return 0;
    RayGen gen;
    if (gen_ptr) gen = *gen_ptr; else memset(&gen, 0, sizeof(gen));
    NM_REQUIRE(samples >= 1, "composite: samples per ray must be >= 1");
    const dim3 grid((unsigned)((rays + 3) / 4)), block(256);
    const int per = (samples + 63) / 64;
    if (per > 8) {                        // more than 512 samples: segments of 512 (composite_long_kernel)
        hipLaunchKernelGGL(composite_long_kernel, grid, block, 0, stream, d_radiance, d_t, d_dirs, d_noise, rays, samples, thr, white_bg,
                           training, out, gen);
        NM_HIP_CHECK(hipGetLastError());
        return 0;
    }
#define NM_COMPOSITE(P)                                                                                      \
    hipLaunchKernelGGL(composite_kernel<P>, grid, block, 0, stream, d_radiance, d_t, d_dirs, d_noise, rays, samples, \
                       thr, white_bg, training, out, gen)
    switch (per) {
        case 1: NM_COMPOSITE(1); break;
        case 2: NM_COMPOSITE(2); break;
        case 3: NM_COMPOSITE(3); break;
        case 4: NM_COMPOSITE(4); break;
        default: NM_COMPOSITE(8); break;
    }
#undef NM_COMPOSITE
    NM_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_sample_pdf(const float* d_t, const float* d_weights, const float* d_u, int u_per_ray, int64_t rays, int coarse,
                      int fine, float* d_t_out, hipStream_t stream) {
    if (rays <= 0) return 0;
    NM_REQUIRE(coarse >= 3 && fine >= 1, "sample_pdf: num_coarse must be >= 3 and num_fine >= 1");
    NM_REQUIRE(3 * (int64_t)coarse + fine <= PDF_MAX_FLOATS_PER_WAVE,
               "sample_pdf: 3 * num_coarse + num_fine must be <= 10240 (a ray's cdf, bins and merged depths live in LDS)");
    const int lds_bytes = 4 * (3 * coarse + fine) * 4;
    if (int rc = ensure_dynamic_lds((const void*)sample_pdf_kernel, lds_bytes)) return rc;
    hipLaunchKernelGGL(sample_pdf_kernel, dim3((unsigned)((rays + 3) / 4)), dim3(256), lds_bytes, stream, d_t, d_weights, d_u,
                       u_per_ray, rays, coarse, fine, d_t_out);
    NM_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace nm

using namespace nm;

extern "C" {

int nm_ray_bundle(const float* h_c2w, int32_t height, int32_t width, float focal, int64_t first, int64_t count,
                  float* d_dirs, float* h_origin, void* stream) {
    NM_REQUIRE(h_c2w && d_dirs && height > 0 && width > 0, "bad argument");
    NM_REQUIRE(first >= 0 && count >= 0 && first + count <= (int64_t)height * width, "pixel range");
    if (h_origin)
        for (int a = 0; a < 3; ++a) h_origin[a] = h_c2w[4 * a + 3];
    return launch_ray_bundle(h_c2w, height, width, focal, first, count, d_dirs, static_cast<hipStream_t>(stream));
}

int nm_coarse_intervals(const float* d_u, const float* d_near, const float* d_far, int bounds_per_ray, int lindisp,
                        int64_t rays, int32_t samples, float* d_t, void* stream) {
    NM_REQUIRE(d_u && d_near && d_far && d_t && samples > 0 && rays >= 0, "bad argument");
    return launch_coarse_intervals(d_u, d_near, d_far, bounds_per_ray, lindisp, rays, samples, d_t,
                                   static_cast<hipStream_t>(stream));
}

int nm_composite(const float* d_radiance, const float* d_t, const float* d_dirs, int64_t rays, int32_t samples,
                 float attenuation_threshold, int white_background, int training, const nm_bundle_out* out,
                 void* stream) {
    NM_REQUIRE(d_radiance && d_t && d_dirs && out && rays >= 0, "bad argument");
    return launch_composite(d_radiance, d_t, d_dirs, nullptr, rays, samples, attenuation_threshold, white_background,
                            training, *out, static_cast<hipStream_t>(stream));
}

int nm_sample_pdf(const float* d_t, const float* d_weights, const float* d_u, int64_t rays, int32_t coarse,
                  int32_t fine, float* d_t_out, void* stream) {
    NM_REQUIRE(d_t && d_weights && d_u && d_t_out && rays >= 0, "bad argument");
    return launch_sample_pdf(d_t, d_weights, d_u, 0, rays, coarse, fine, d_t_out, static_cast<hipStream_t>(stream));
}

int nm_perturb_intervals(const float* d_t, const float* d_rand, int64_t rays, int32_t samples, float* d_t_out,
                         void* stream) {
    NM_REQUIRE(d_t && d_rand && d_t_out && rays >= 0 && samples > 0, "bad argument");
    const int64_t n = rays * samples;
    if (n == 0) return 0;
    hipLaunchKernelGGL(perturb_intervals_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), d_t, d_rand, rays, samples, d_t_out);
    NM_HIP_CHECK(hipGetLastError());
    return 0;
}

int nm_composite_train(const float* d_radiance, const float* d_t, const float* d_dirs, const float* d_noise,
                       int64_t rays, int32_t samples, float attenuation_threshold, int white_background,
                       const nm_bundle_out* out, void* stream) {
    NM_REQUIRE(d_radiance && d_t && d_dirs && out && rays >= 0, "bad argument");
    return launch_composite(d_radiance, d_t, d_dirs, d_noise, rays, samples, attenuation_threshold, white_background, 1,
                            *out, static_cast<hipStream_t>(stream));
}

int nm_composite_backward(const float* d_radiance, const float* d_t, const float* d_dirs, const float* d_noise,
                          int64_t rays, int32_t samples, int white_background, const nm_bundle_grads* grads,
                          float* d_grad_radiance, void* stream) {
    NM_REQUIRE(d_radiance && d_t && d_dirs && grads && d_grad_radiance && rays >= 0, "bad argument");
    NM_REQUIRE(samples >= 1, "composite: samples per ray must be >= 1");
    if (rays == 0) return 0;
    const dim3 grid((unsigned)((rays + 3) / 4)), block(256);
    if (samples > 512) {
        hipLaunchKernelGGL(composite_backward_long_kernel, grid, block, 0, static_cast<hipStream_t>(stream), d_radiance, d_t, d_dirs,
                           d_noise, rays, samples, white_background, *grads, d_grad_radiance);
        NM_HIP_CHECK(hipGetLastError());
        return 0;
    }
#define NM_COMPOSITE_BWD(P)                                                                                                     \
    hipLaunchKernelGGL(composite_backward_kernel<P>, grid, block, 0, static_cast<hipStream_t>(stream), d_radiance, d_t, d_dirs, \
                       d_noise, rays, samples, white_background, *grads, d_grad_radiance)
    switch ((samples + 63) / 64) {
        case 1: NM_COMPOSITE_BWD(1); break;
        case 2: NM_COMPOSITE_BWD(2); break;
        case 3: NM_COMPOSITE_BWD(3); break;
        case 4: NM_COMPOSITE_BWD(4); break;
        default: NM_COMPOSITE_BWD(8); break;
    }
#undef NM_COMPOSITE_BWD
    NM_HIP_CHECK(hipGetLastError());
    return 0;
}

int nm_sample_pdf_rand(const float* d_t, const float* d_weights, const float* d_u, int64_t rays, int32_t coarse,
                       int32_t fine, float* d_t_out, void* stream) {
    NM_REQUIRE(d_t && d_weights && d_u && d_t_out && rays >= 0, "bad argument");
    return launch_sample_pdf(d_t, d_weights, d_u, 1, rays, coarse, fine, d_t_out, static_cast<hipStream_t>(stream));
}

static inline size_t align256(size_t x) { return (x + 255) & ~size_t(255); }

int64_t nm_render_workspace_bytes(int64_t rays, int32_t num_coarse, int32_t num_fine) {
    const size_t sc = (size_t)num_coarse, sf = (size_t)(num_coarse + (num_fine > 0 ? num_fine : 0));
    size_t b = 0;
    b += align256(rays * sc * 4);        // t coarse
    b += align256(rays * sc * 16);       // radiance coarse
    b += align256(rays * sc * 4);        // weights coarse (when the caller does not ask for them)
    if (num_fine > 0) {
        b += align256(rays * sf * 4);    // t fine
        b += align256(rays * sf * 16);   // radiance fine
    }
    return (int64_t)b;
}

// NeRFModel.forward (model_nerf.py:37-78); `gen` != nullptr: the rays come from the pose (no origin / direction buffers)
static int render_impl(nm_mlp* coarse, nm_mlp* fine, const nm_render_cfg* cfg, const float* d_origins, int origins_per_ray,
                       const float* d_dirs, const RayGen* gen, const float* d_near, const float* d_far,
                       int bounds_per_ray, const float* d_u_coarse, const float* d_u_fine, int64_t rays,
                       void* d_workspace, const nm_bundle_out* coarse_out, const nm_bundle_out* fine_out,
                       hipStream_t stream) {
    if (rays <= 0) return 0;
    const int sc = cfg->num_coarse, sf = cfg->num_coarse + cfg->num_fine;
    char* ws = static_cast<char*>(d_workspace);
    float* t_c = reinterpret_cast<float*>(ws); ws += align256(rays * (size_t)sc * 4);
    float* rad_c = reinterpret_cast<float*>(ws); ws += align256(rays * (size_t)sc * 16);
    float* w_c = reinterpret_cast<float*>(ws); ws += align256(rays * (size_t)sc * 4);
    int rc;
    auto mlp = [&](nm_mlp* m, const float* t, int samples, float* rad) -> int {
        if (!gen) return nm_mlp_eval_rays(m, d_origins, origins_per_ray, d_dirs, t, rays, samples, rad, stream);
        return nm_mlp_eval_view_internal(m, gen, t, rays, samples, rad, stream);
    };
    // RaySampleInterval -> intervals_to_ray_points -> model_coarse -> volume_renderer (model_nerf.py:52-62)
    if ((rc = launch_coarse_intervals(d_u_coarse, d_near, d_far, bounds_per_ray, cfg->lindisp, rays, sc, t_c, stream))) return rc;
    if ((rc = mlp(coarse, t_c, sc, rad_c))) return rc;
    nm_bundle_out co = *coarse_out;
    if (!co.d_weights) co.d_weights = w_c;
    if ((rc = launch_composite(rad_c, t_c, d_dirs, nullptr, rays, sc, cfg->attenuation_threshold, cfg->white_background,
                               cfg->training, co, stream, gen))) return rc;
    if (!fine) return 0;
    // sample_pdf -> intervals_to_ray_points -> model_fine -> volume_renderer (model_nerf.py:65-76)
    float* t_f = reinterpret_cast<float*>(ws); ws += align256(rays * (size_t)sf * 4);
    float* rad_f = reinterpret_cast<float*>(ws);
    if ((rc = launch_sample_pdf(t_c, co.d_weights, d_u_fine, 0, rays, sc, cfg->num_fine, t_f, stream))) return rc;
    if ((rc = mlp(fine, t_f, sf, rad_f))) return rc;
    return launch_composite(rad_f, t_f, d_dirs, nullptr, rays, sf, cfg->attenuation_threshold, cfg->white_background,
                            cfg->training, *fine_out, stream, gen);
}

int nm_render_rays(nm_mlp* coarse, nm_mlp* fine, const nm_render_cfg* cfg, const float* d_origins, int origins_per_ray,
                   const float* d_dirs, const float* d_near, const float* d_far, int bounds_per_ray,
                   const float* d_u_coarse, const float* d_u_fine, int64_t rays, void* d_workspace,
                   const nm_bundle_out* coarse_out, const nm_bundle_out* fine_out, void* stream_) {
    NM_REQUIRE(coarse && cfg && d_origins && d_dirs && d_near && d_far && d_u_coarse && d_workspace && coarse_out,
               "bad argument");
    NM_REQUIRE(!fine || (d_u_fine && fine_out && cfg->num_fine > 0), "fine network needs u_fine / fine_out / num_fine");
    return render_impl(coarse, fine, cfg, d_origins, origins_per_ray, d_dirs, nullptr, d_near, d_far, bounds_per_ray,
                       d_u_coarse, d_u_fine, rays, d_workspace, coarse_out, fine_out, static_cast<hipStream_t>(stream_));
}

static int make_raygen(const nm_view* v, int64_t first, RayGen* g) {
    NM_REQUIRE(v && v->height > 0 && v->width > 0 && v->focal > 0.0, "bad view");
    memset(g, 0, sizeof(*g));
    for (int a = 0; a < 3; ++a) {
        for (int b = 0; b < 3; ++b) g->rot[3 * a + b] = v->c2w[4 * a + b];
        g->origin[a] = v->c2w[4 * a + 3];
    }
    g->height = v->height; g->width = v->width; g->focal = (float)v->focal;
    g->enabled = 1; g->ndc = v->use_ndc ? 1 : 0; g->first = first;
    // ndc_rays' python-float constants (nerf_helpers.py:288-303), computed in fp64 and rounded once, as torch does
    g->ndc_near = (float)v->ndc_near;
    g->c_w = (float)(-1.0 / (v->width / (2.0 * v->focal)));
    g->c_h = (float)(-1.0 / (v->height / (2.0 * v->focal)));
    g->two_near = (float)(2.0 * v->ndc_near);
    g->m_two_near = (float)(-2.0 * v->ndc_near);
    return 0;
}

int nm_render_view(nm_mlp* coarse, nm_mlp* fine, const nm_render_cfg* cfg, const nm_view* view, int64_t first_pixel,
                   int64_t rays, const float* d_near, const float* d_far, int bounds_per_ray, const float* d_u_coarse,
                   const float* d_u_fine, void* d_workspace, const nm_bundle_out* coarse_out,
                   const nm_bundle_out* fine_out, void* stream_) {
    NM_REQUIRE(coarse && cfg && view && d_near && d_far && d_u_coarse && d_workspace && coarse_out, "bad argument");
    NM_REQUIRE(!fine || (d_u_fine && fine_out && cfg->num_fine > 0), "fine network needs u_fine / fine_out / num_fine");
    NM_REQUIRE(first_pixel >= 0 && rays >= 0 && first_pixel + rays <= (int64_t)view->height * view->width, "pixel range");
    RayGen g;
    int rc = make_raygen(view, first_pixel, &g);
    if (rc) return rc;
    return render_impl(coarse, fine, cfg, nullptr, 0, nullptr, &g, d_near, d_far, bounds_per_ray, d_u_coarse, d_u_fine,
                       rays, d_workspace, coarse_out, fine_out, static_cast<hipStream_t>(stream_));
}

int nm_view_rays(const nm_view* view, int64_t first_pixel, int64_t rays, float* d_origins, float* d_dirs, void* stream_) {
    NM_REQUIRE(view && d_origins && d_dirs, "bad argument");
    NM_REQUIRE(first_pixel >= 0 && rays >= 0 && first_pixel + rays <= (int64_t)view->height * view->width, "pixel range");
    RayGen g;
    int rc = make_raygen(view, first_pixel, &g);
    if (rc || rays == 0) return rc;
    hipLaunchKernelGGL(view_rays_kernel, dim3((unsigned)((rays + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream_), g, rays, d_origins, d_dirs);
    NM_HIP_CHECK(hipGetLastError());
    return 0;
}

int nm_ndc_rays(int32_t height, int32_t width, double focal, double near_, const float* d_origins, int origins_per_ray,
                const float* d_dirs, int64_t n, float* d_out_origins, float* d_out_dirs, void* stream_) {
    NM_REQUIRE(d_origins && d_dirs && d_out_origins && d_out_dirs && height > 0 && width > 0 && focal > 0.0 && n >= 0,
               "bad argument");
    if (n == 0) return 0;
    hipLaunchKernelGGL(ndc_rays_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream_),
                       d_origins, origins_per_ray, d_dirs, n, (float)near_, (float)(-1.0 / (width / (2.0 * focal))),
                       (float)(-1.0 / (height / (2.0 * focal))), (float)(2.0 * near_), (float)(-2.0 * near_),
                       d_out_origins, d_out_dirs);
    NM_HIP_CHECK(hipGetLastError());
    return 0;
}

int nm_positional_encoding(const float* d_x, int64_t n, int32_t dim, const float* h_bands, int32_t num_bands,
                           int32_t include_input, float* d_out, void* stream_) {
    NM_REQUIRE(d_x && h_bands && d_out && n >= 0 && dim >= 1, "bad argument");
    NM_REQUIRE(num_bands >= 0 && num_bands <= MAX_FREQ_XYZ, "positional_encoding: at most 16 frequency bands");
    if (n == 0) return 0;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int width = 2 * dim * num_bands + (include_input ? dim : 0);
    if (num_bands == 0) {   // encoding = the input itself (or nothing)
        if (include_input) NM_HIP_CHECK(hipMemcpyAsync(d_out, d_x, (size_t)n * dim * 4, hipMemcpyDeviceToDevice, stream));
        (void)width;
        return 0;
    }
    BandArgs b;
    for (int k = 0; k < MAX_FREQ_XYZ; ++k) b.f[k] = k < num_bands ? h_bands[k] : 0.0f;
    const int64_t work = n * (int64_t)(dim * num_bands);
    hipLaunchKernelGGL(positional_encoding_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, stream, d_x, n,
                       dim, num_bands, include_input, b, d_out);
    NM_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // extern "C"
