// What bounds the opt-in bf16x3 mode, and would 32 x 32 sample tiles lift it?     
//   hipcc --offload-arch=gfx950 -O3 -o tests/tools/probes/b3_tiles tests/tools/probes/b3_tiles.hip && tests/tools/probes/b3_tiles
// The mode replaces every fp32 product by six bf16 MFMAs of a three-way split of both operands.  Per (k-block, output tile) "unit"
// a wave reads the three weight planes from LDS (3 x ds_read_b128 = 3 KiB per wave) and issues six MFMAs against activation planes
// it holds in registers.  With v_mfma_f32_16x16x32_bf16 (16 samples per wave; what mlp_device_b3.h uses) that is 3 KiB of LDS per
// 96 matrix cycles and SIMD: 128 B / clk per CU at the full rate -- exactly the LDS peak.  v_mfma_f32_32x32x16_bf16 multiplies the
// same 3 KiB against 32 samples: half the LDS bytes per FLOP, at four times the accumulator registers per output tile.
// This probe runs ONLY that inner loop (no encodings, no bias / ReLU / split, no weight DMA: the planes sit in LDS) for both tile
// shapes, with the operand reads and without them, at one and two waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int UNITS = 16;              // units resident in LDS: 16 x 3 KiB = 48 KiB, walked round and round

template <bool WIDE, bool READ>
__global__ __launch_bounds__(512) void b3_loop(const u32x4* __restrict__ planes, int iters, float* out) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < UNITS * 3 * 64; i += blockDim.x) reinterpret_cast<u32x4*>(lds)[i] = planes[i];
    __syncthreads();
    const bf16x8 x1 = __builtin_bit_cast(bf16x8, planes[lane]), x2 = __builtin_bit_cast(bf16x8, planes[64 + lane]),
                 x3 = __builtin_bit_cast(bf16x8, planes[128 + lane]);
    u32x4 a[3] = {planes[lane], planes[64 + lane], planes[128 + lane]};
    constexpr int NT = WIDE ? 8 : 16;                                // output tiles of a 256-wide layer
    typedef typename std::conditional<WIDE, f32x16, f32x4>::type acc_t;
    acc_t acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = acc_t{};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < UNITS; ++u) {
            if (READ) {
#pragma unroll
                for (int p = 0; p < 3; ++p) a[p] = *reinterpret_cast<const u32x4*>(lds + (u * 3 + p) * 1024 + lane * 16);
            }
            const bf16x8 a1 = __builtin_bit_cast(bf16x8, a[0]), a2 = __builtin_bit_cast(bf16x8, a[1]), a3 = __builtin_bit_cast(bf16x8, a[2]);
            acc_t d = acc[u % NT];
            if constexpr (WIDE) {
                d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, x1, d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, x2, d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, x3, d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, x1, d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, x2, d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, x1, d, 0, 0, 0);
            } else {
                d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3, x1, d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, x2, d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, x3, d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, x1, d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, x2, d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, x1, d, 0, 0, 0);
            }
            acc[u % NT] = d;
        }
    }
    float r = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) r += acc[t][0] + acc[t][3];
    if (r == 12345.678f) out[0] = r;
}

int main() {
    u32x4* planes; float* out;
    const int n = UNITS * 3 * 64;
    CK(hipMalloc(&planes, n * 16)); CK(hipMalloc(&out, 64));
    unsigned* h = (unsigned*)malloc(n * 16);
    for (int i = 0; i < n * 4; ++i) h[i] = 0x3c003c00u + ((i * 2654435761u) >> 9 & 0x007f007fu);   // bf16 pairs around 0.0078: finite, non-trivial bits
    CK(hipMemcpy(planes, h, n * 16, hipMemcpyHostToDevice));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 2000;
    printf("CUs %d; bf16 dense peak 2500 TFLOP/s -> 416.7 fp32-equivalent at 6 MFMAs per product\n", cus);
    printf("%-34s %-10s %-16s %10s %14s %12s\n", "tile", "operands", "waves per SIMD", "ms", "bf16 TFLOP/s", "fp32-equiv");
#define RUN(WIDE, READ, WAVES) do { \
        const int threads = 256 * (WAVES), wgs = cus; \
        const int ldsb = UNITS * 3 * 1024; \
        auto k = b3_loop<WIDE, READ>; \
        CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb)); \
        for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k, dim3(wgs), dim3(threads), ldsb, 0, planes, iters, out); \
        float best = 1e9f; \
        for (int r = 0; r < 5; ++r) { CK(hipEventRecord(e0)); hipLaunchKernelGGL(k, dim3(wgs), dim3(threads), ldsb, 0, planes, iters, out); \
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; } \
        CK(hipGetLastError()); \
        const double flop = (double)wgs * (threads / 64) * iters * UNITS * 6.0 * (WIDE ? 32768.0 : 16384.0); \
        printf("%-34s %-10s %-16d %10.3f %14.1f %12.1f\n", WIDE ? "32x32x16 (32 samples per wave)" : "16x16x32 (16 samples per wave)", \
               READ ? "LDS" : "registers", WAVES, best, flop / (best * 1e-3) / 1e12, flop / (best * 1e-3) / 1e12 / 6.0); } while (0)
    RUN(false, false, 1); RUN(false, true, 1); RUN(false, false, 2); RUN(false, true, 2);
    RUN(true, false, 1); RUN(true, true, 1); RUN(true, false, 2); RUN(true, true, 2);
    return 0;
}
