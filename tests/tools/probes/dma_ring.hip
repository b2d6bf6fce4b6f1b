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

template <int CHUNK, int SLOTS, bool LOAD>
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
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float sum = 0.f;
    for (int64_t c = 0; c < chunks_per_wg; ++c) {
        dma(c + SLOTS - 1);
        const float a = *reinterpret_cast<const float*>(lds + (int)(c % SLOTS) * CHUNK + lane * 4 + wave * 1024);
        sum += a;
        for (int m = 0; m < mfmas; m += 4) {
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, 1.0f, acc[q], 0, 0, 0);
        }
        if (c + SLOTS - 1 < chunks_per_wg) wait_vm<(SLOTS - 2) * P>();
        else wait_vm<0>();
        __builtin_amdgcn_s_barrier();
    }
    const float r = sum + acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    if (r == 12345.678f) out[0] = r;
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
    return 0;
}
