// How fast does the weight-gradient kernels' dataflow draw from HBM, and does it overlap with the matrix pipe?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/dma_ring tests/tools/probes/dma_ring.hip && /tmp/dma_ring
// One persistent workgroup of 8 waves per slot of a CU (grid = CUs * wgs_per_cu) streams a contiguous range of a 4 GiB
// buffer HBM -> LDS through a ring of SLOTS chunks with scalar-addressed `buffer_load ... lds` pieces (1 KiB per wave
// instruction), SLOTS-1 chunks ahead, a counted s_waitcnt + one bare barrier per chunk (nerf_dw.hip's schedule), and runs
// `mfmas` v_mfma_f32_16x16x4_f32 per wave and chunk on the side.  Printed per configuration: time, TB/s, and the time the
// MFMAs alone take (bytes = 0) -- max(t_hbm, t_mfma) would be perfect overlap, their sum none.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int CHUNK, int SLOTS, bool LOAD, bool OPS = false>
__global__ __launch_bounds__(512) void ring(const char* __restrict__ src, int64_t chunks_per_wg, int mfmas, float* out) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int NW = 8, P = CHUNK / 1024 / NW;
    static_assert(P >= 1 && (SLOTS - 2) * P <= 60, "pieces");
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const char* base = src + (int64_t)blockIdx.x * chunks_per_wg * CHUNK;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), (short)16, 0x7fffffff, 1 << 23);
    auto dma = [&](int64_t c) {
        if (!LOAD || c >= chunks_per_wg) return;
        char* dst = lds + (int)(c % SLOTS) * CHUNK;
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const int piece = wave + p * NW;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(dst + piece * 1024), 16, 0,
                                                     (int)(c * CHUNK) + piece * 1024, 0, 0);
        }
    };
    for (int c = 0; c < SLOTS - 1; ++c) dma(c);
    wait_vm<(SLOTS - 2) * P>();
    __syncthreads();
    f32x4 acc[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = f32x4{0, 0, 0, 0};
    float sum = 0.f;
    for (int64_t c = 0; c < chunks_per_wg; ++c) {
        dma(c + SLOTS - 1);
        const char* slot = lds + (int)(c % SLOTS) * CHUNK;
        const float a = *reinterpret_cast<const float*>(slot + lane * 4 + wave * 1024);
        sum += a;
        if (OPS) {
            // the weight-gradient kernels' operand traffic: two ds_read_b128 per 16 MFMAs, operands straight from the chunk
            for (int m = 0; m < mfmas; m += 16) {
                const f32x4 x = *reinterpret_cast<const f32x4*>(slot + ((m * 64 + lane * 16) & (CHUNK / 2 - 1)));
                const f32x4 y = *reinterpret_cast<const f32x4*>(slot + CHUNK / 2 + ((m * 64 + lane * 16) & (CHUNK / 2 - 1)));
#pragma unroll
                for (int qa = 0; qa < 4; ++qa)
#pragma unroll
                    for (int qb = 0; qb < 4; ++qb) acc[qa * 4 + qb] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[qa], y[qb], acc[qa * 4 + qb], 0, 0, 0);
            }
        } else {
            for (int m = 0; m < mfmas; m += 4) {
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, 1.0f, acc[q], 0, 0, 0);
            }
        }
        if (c + SLOTS - 1 < chunks_per_wg) wait_vm<(SLOTS - 2) * P>();
        else wait_vm<0>();
        __builtin_amdgcn_s_barrier();
    }
    float r = sum;
#pragma unroll
    for (int q = 0; q < 16; ++q) r += acc[q][q & 3];
    if (r == 12345.678f) out[0] = r;
}

__global__ void fill_random(float* p, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t)(i * 2654435761u) ^ (uint32_t)(i >> 32);
        h ^= h >> 15; h *= 0x2c1b3c6du; h ^= h >> 12; h *= 0x297a2d39u; h ^= h >> 15;
        p[i] = ((h >> 8) * (1.0f / 8388608.0f) - 1.0f) * ((h & 1) ? 1.0f : 0.0f);   // ~half zeros (ReLU rows), the rest uniform in [-1, 1)
    }
}

int main() {
    const int64_t bytes = 4LL << 30;
    char* d; float* out;
    CK(hipMalloc(&d, bytes + (1 << 20))); CK(hipMalloc(&out, 64));
    CK(hipMemset(d, 0, bytes));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto time = [&](auto launch) {
        for (int w = 0; w < 2; ++w) launch();
        float best = 1e9f;
        for (int r = 0; r < 5; ++r) {
            CK(hipEventRecord(a)); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
        }
        CK(hipGetLastError());
        return best;
    };
    printf("CUs %d\nchunk_KiB slots wgs_per_cu in_flight_KiB_per_cu mfmas_per_wave_chunk  ms_both  TB/s  ms_mfma_only  ms_load_only\n", cus);
#define RUN(CH, SL, WPC, MF) do { \
        const int grid = cus * (WPC); const int64_t cpw = bytes / ((int64_t)grid * (CH)); const int ldsb = (CH) * (SL); \
        CK(hipFuncSetAttribute((const void*)ring<CH, SL, true>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb)); \
        CK(hipFuncSetAttribute((const void*)ring<CH, SL, false>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb)); \
        const float both = time([&] { hipLaunchKernelGGL((ring<CH, SL, true>), dim3(grid), dim3(512), ldsb, 0, d, cpw, MF, out); }); \
        const float mf = (MF) ? time([&] { hipLaunchKernelGGL((ring<CH, SL, false>), dim3(grid), dim3(512), ldsb, 0, d, cpw, MF, out); }) : 0.f; \
        const float ld = (MF) ? time([&] { hipLaunchKernelGGL((ring<CH, SL, true>), dim3(grid), dim3(512), ldsb, 0, d, cpw, 0, out); }) : both; \
        printf("%4d %3d %2d %5d %4d  %8.3f %6.2f %8.3f %8.3f\n", (CH) / 1024, SL, WPC, ((SL) - 1) * ((CH) / 1024) * (WPC), MF, both, \
               (double)cpw * grid * (CH) / (both * 1e-3) / 1e12, mf, ld); } while (0)
    // pure streaming: in-flight depth and workgroups per CU
    RUN(8192, 4, 1, 0); RUN(16384, 4, 1, 0); RUN(32768, 4, 1, 0); RUN(16384, 8, 1, 0); RUN(8192, 16, 1, 0);
    RUN(8192, 4, 2, 0); RUN(16384, 4, 2, 0); RUN(32768, 2, 2, 0); RUN(8192, 4, 4, 0);
    RUN(8192, 8, 2, 0); RUN(16384, 3, 3, 0);
    // with the matrix pipe busy: 128x128 products (16 KiB per 16 rows: 32 MFMAs per wave), 64x64 (8 KiB: 8 MFMAs per wave and 16 rows),
    // 256x256 (32 KiB: 128)
    RUN(16384, 4, 1, 32); RUN(16384, 8, 1, 32); RUN(16384, 4, 2, 32); RUN(32768, 4, 1, 64);
    RUN(8192, 4, 1, 8); RUN(8192, 16, 1, 8); RUN(32768, 4, 1, 32); RUN(8192, 4, 4, 8);
    RUN(32768, 4, 1, 128); RUN(32768, 2, 2, 128);
    // the same with REAL operands: ds_read_b128 pairs feeding 16 independent accumulators, first on the zero buffer, then on
    // random data (half zeros like ReLU rows): the matrix pipe's power draw -- hence the clock -- depends on the bits it multiplies
#define RUNOPS(CH, SL, WPC, MF, TAG) do { \
        const int grid = cus * (WPC); const int64_t cpw = bytes / ((int64_t)grid * (CH)); const int ldsb = (CH) * (SL); \
        CK(hipFuncSetAttribute((const void*)ring<CH, SL, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb)); \
        CK(hipFuncSetAttribute((const void*)ring<CH, SL, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb)); \
        const float both = time([&] { hipLaunchKernelGGL((ring<CH, SL, true, true>), dim3(grid), dim3(512), ldsb, 0, d, cpw, MF, out); }); \
        const float mf = time([&] { hipLaunchKernelGGL((ring<CH, SL, false, true>), dim3(grid), dim3(512), ldsb, 0, d, cpw, MF, out); }); \
        const double flop = (double)cpw * grid * 8 * (MF) * 2048.0; \
        printf("%-7s %4d %3d %2d %4d  both %8.3f ms %6.2f TB/s %6.1f TFLOP/s | mfma on stale LDS only %8.3f ms %6.1f TFLOP/s\n", TAG, (CH) / 1024, SL, WPC, MF, both, \
               (double)cpw * grid * (CH) / (both * 1e-3) / 1e12, flop / (both * 1e-3) / 1e12, mf, flop / (mf * 1e-3) / 1e12); } while (0)
    printf("data   chunk_KiB slots wgs_per_cu mfmas_per_wave_chunk\n");
    for (int pass = 0; pass < 2; ++pass) {
        const char* tag = pass ? "random" : "zeros";
        if (pass) { hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, (float*)d, bytes / 4); CK(hipDeviceSynchronize()); }
        RUNOPS(16384, 4, 1, 32, tag); RUNOPS(16384, 4, 2, 32, tag); RUNOPS(32768, 4, 1, 64, tag); RUNOPS(32768, 4, 1, 128, tag); RUNOPS(8192, 4, 1, 16, tag);
        RUNOPS(8192, 4, 2, 16, tag); RUNOPS(32768, 4, 1, 32, tag);
    }
    return 0;
}
