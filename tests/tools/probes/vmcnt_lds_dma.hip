// How many vmcnt units does one `buffer_load_dwordx4 ... lds` (wave64, 1 KiB) take?  Issue P pieces from cold HBM addresses, then
// s_waitcnt vmcnt(N): the wait returns at once iff N >= the units those pieces hold.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i32x4 rsrc_of(const void* base) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(base);
    return i32x4{(int)(unsigned)a, (int)(((unsigned)(a >> 32) & 0xffffu) | (16u << 16)), 0x7fffffff, 1 << 23};
}
template <int P, int N>
__global__ void k(const char* src, long long* out, size_t stride) {
    extern __shared__ char lds[];
    const i32x4 r = rsrc_of(src + (size_t)blockIdx.x * stride);
    const unsigned d = (unsigned)reinterpret_cast<size_t>((__attribute__((address_space(3))) char*)lds);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int p = 0; p < P; ++p)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 off, %1, %2 lds" ::"s"(d + p * 1024), "s"(r), "s"(p * 4096) : "memory");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
    const long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t2 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = t2 - t0; }
}
template <int P, int N> void run(const char* src, long long* out, int rep) {
    long long h[2];
    long long a = 0, b = 0;
    for (int i = 0; i < 8; ++i) {
        hipLaunchKernelGGL((k<P, N>), dim3(1), dim3(64), 16384, 0, src + (size_t)(rep * 8 + i) * (64u << 20), out, (size_t)0);
        CK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost));
        a += h[0]; b += h[1];
    }
    printf("  %d pieces, s_waitcnt vmcnt(%d): %5.0f ticks;  then vmcnt(0): %5.0f ticks (100 MHz ticks, mean of 8 cold launches)\n", P, N, a / 8.0, b / 8.0);
}
int main() {
    char* src; CK(hipMalloc(&src, (size_t)12 << 30)); CK(hipMemset(src, 1, (size_t)12 << 30));
    long long* out; CK(hipMalloc(&out, 64));
    run<1, 0>(src, out, 0); run<1, 1>(src, out, 1); run<2, 1>(src, out, 2); run<2, 2>(src, out, 3); run<4, 3>(src, out, 4); run<4, 4>(src, out, 5);
    run<4, 8>(src, out, 6); run<4, 16>(src, out, 7); run<8, 8>(src, out, 8); run<8, 7>(src, out, 9); run<8, 32>(src, out, 10);
    return 0;
}
