// Read-bandwidth probe for the marching-cubes classify pass: how fast can 480^3 floats (442 MB) be streamed from HBM
// by kernels shaped like mc_classify_stream, with the classification removed?   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// flat grid-stride: T threads per workgroup, U 16-byte loads in flight per thread
template <int T, int U>
__global__ __launch_bounds__(T) void flat(const float4* __restrict__ p, int64_t n4, float* out) {
    float acc = 0.f;
    const int64_t stride = (int64_t)gridDim.x * T;
    int64_t i = (int64_t)blockIdx.x * T + threadIdx.x;
    for (; i + (U - 1) * stride < n4; i += U * stride) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = p[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    for (; i < n4; i += stride) { const float4 v = p[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 12345.678f) out[0] = acc;
}

// contiguous slab per workgroup: WG b reads bytes [b * slab, (b+1) * slab) in steps of T * 16 B, AHEAD steps in flight,
// optional barrier per step (the classify kernel's shape)
template <int T, int AHEAD, bool BARRIER>
__global__ __launch_bounds__(T) void slab(const float4* __restrict__ p, int64_t n4, int steps, float* out) {
    const int64_t base = (int64_t)blockIdx.x * steps * T + threadIdx.x;
    float acc = 0.f;
    float4 v[AHEAD];
#pragma unroll
    for (int a = 0; a < AHEAD; ++a) { int64_t i = base + (int64_t)a * T; v[a] = p[i < n4 ? i : n4 - 1]; }
    for (int s0 = 0; s0 < steps; s0 += AHEAD) {
#pragma unroll
        for (int a = 0; a < AHEAD; ++a) {
            const float4 q = v[a];
            int64_t i = base + (int64_t)(s0 + a + AHEAD) * T;
            v[a] = p[(s0 + a + AHEAD < steps && i < n4) ? i : base];
            acc += q.x + q.y + q.z + q.w;
            if (BARRIER) __syncthreads();
        }
    }
    if (acc == 12345.678f) out[0] = acc;
}

int main() {
    const int64_t n = 480LL * 480 * 480, n4 = n / 4;
    float4* d; float* out;
    CK(hipMalloc(&d, n * 4 + 4096)); CK(hipMalloc(&out, 64));
    CK(hipMemset(d, 0, n * 4));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto time = [&](const char* name, auto launch) {
        for (int w = 0; w < 3; ++w) launch();
        float best = 1e9f;
        for (int r = 0; r < 10; ++r) {
            CK(hipEventRecord(a)); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
        }
        printf("%-58s %7.1f us  %6.2f TB/s\n", name, best * 1e3, n * 4 / (best * 1e-3) / 1e12);
    };
    for (int wgs : {256 * 2, 256 * 4, 256 * 8, 256 * 16, 256 * 32}) {
        char nm[128];
        snprintf(nm, sizeof nm, "flat<256,4> grid %d", wgs); time(nm, [&] { hipLaunchKernelGGL((flat<256, 4>), dim3(wgs), dim3(256), 0, 0, d, n4, out); });
        snprintf(nm, sizeof nm, "flat<256,8> grid %d", wgs); time(nm, [&] { hipLaunchKernelGGL((flat<256, 8>), dim3(wgs), dim3(256), 0, 0, d, n4, out); });
    }
    for (int wgs : {256, 512, 1024}) {
        char nm[128];
        snprintf(nm, sizeof nm, "flat<1024,4> grid %d", wgs); time(nm, [&] { hipLaunchKernelGGL((flat<1024, 4>), dim3(wgs), dim3(1024), 0, 0, d, n4, out); });
        snprintf(nm, sizeof nm, "flat<1024,8> grid %d", wgs); time(nm, [&] { hipLaunchKernelGGL((flat<1024, 8>), dim3(wgs), dim3(1024), 0, 0, d, n4, out); });
    }
    for (int steps : {24, 48, 96}) {
        char nm[128];
        const int g1024 = (int)((n4 + (int64_t)steps * 1024 - 1) / ((int64_t)steps * 1024));
        const int g256 = (int)((n4 + (int64_t)steps * 256 - 1) / ((int64_t)steps * 256));
        snprintf(nm, sizeof nm, "slab<1024,6,barrier> steps %d grid %d", steps, g1024);
        time(nm, [&] { hipLaunchKernelGGL((slab<1024, 6, true>), dim3(g1024), dim3(1024), 0, 0, d, n4, steps, out); });
        snprintf(nm, sizeof nm, "slab<1024,6,free> steps %d grid %d", steps, g1024);
        time(nm, [&] { hipLaunchKernelGGL((slab<1024, 6, false>), dim3(g1024), dim3(1024), 0, 0, d, n4, steps, out); });
        snprintf(nm, sizeof nm, "slab<256,6,barrier> steps %d grid %d", steps, g256);
        time(nm, [&] { hipLaunchKernelGGL((slab<256, 6, true>), dim3(g256), dim3(256), 0, 0, d, n4, steps, out); });
        snprintf(nm, sizeof nm, "slab<256,6,free> steps %d grid %d", steps, g256);
        time(nm, [&] { hipLaunchKernelGGL((slab<256, 6, false>), dim3(g256), dim3(256), 0, 0, d, n4, steps, out); });
        snprintf(nm, sizeof nm, "slab<256,12,free> steps %d grid %d", steps, g256);
        time(nm, [&] { hipLaunchKernelGGL((slab<256, 12, false>), dim3(g256), dim3(256), 0, 0, d, n4, steps, out); });
    }
    return 0;
}
