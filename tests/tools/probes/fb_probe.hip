// Where does the fused 64-wide backward kernel (nerfmeshes_amd/csrc/nerf_bwd_fused.hip) spend its time?  The kernel itself, included
// as it is, on synthetic operands of config 1's size (262 144 samples, 4 layers, skip at 2), with parts of it compiled out:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o tests/tools/probes/fb_probe tests/tools/probes/fb_probe.hip && tests/tools/probes/fb_probe
// ABL bits: 1 no dW products (reads + MFMAs), 2 no activation-row DMA, 4 no chain MFMAs, 8 no delta tile writes, 16 / 32 no B / A operand reads in the dW products, 64 no barrier between a delta's
// two phases, 128 no waits and no barriers at all.  Results of the
// ablated variants are wrong on purpose; only the times mean something.
#define NM_FB_KERNEL_ONLY
#include "../../../nerfmeshes_amd/csrc/nerf_bwd_fused.hip"
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
namespace nm { void set_error(const std::string&) {} }
using namespace nm;

template <typename T> T* dev_random(size_t count, unsigned seed, float lo, float hi) {
    std::vector<T> h(count);
    unsigned s = seed * 2654435761u + 12345u;
    for (size_t i = 0; i < count; ++i) { s = s * 1664525u + 1013904223u; h[i] = (T)(lo + (hi - lo) * ((s >> 8) * (1.0f / 16777216.0f))); }
    T* d; CK(hipMalloc(&d, count * sizeof(T))); CK(hipMemcpy(d, h.data(), count * sizeof(T), hipMemcpyHostToDevice));
    return d;
}

template <int MAXL, int ABL>
float run(const MlpBwdArgs& a, const FusedBwdArgs& fa, int L, int grid, int reps) {
    auto k = &mlp_backward_dw64_kernel<MAXL, ABL>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, FB_LDS));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(512), FB_LDS, 0, a, fa, L);
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(512), FB_LDS, 0, a, fa, L);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps * 1e3f;
}

int main(int argc, char** argv) {
    const int L = argc > 1 ? atoi(argv[1]) : 4;
    const int64_t n = argc > 2 ? atoll(argv[2]) : 262144;
    const int sk = L == 4 ? 2 : (L == 8 ? 4 : -1);
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    MlpBwdArgs a{};
    a.wstream = (const char*)dev_random<float>((size_t)(2 * L + 1) * 2048, 1, -0.1f, 0.1f);
    a.walpha = dev_random<float>(64, 2, -0.1f, 0.1f);
    a.wrgb = dev_random<float>(96, 3, -0.1f, 0.1f);
    a.radiance = dev_random<float>(4 * n, 4, 0.05f, 0.95f);
    a.grad_out = dev_random<float>(4 * n, 5, -1.f, 1.f);
    a.n = n; a.tiles = n / 16;
    // ReLU masks: random bits (about half of the activations alive)
    {
        std::vector<uint64_t> h((size_t)(L + 1) * a.tiles * 64);
        unsigned long long s = 88172645463325252ull;
        for (auto& v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = s; }
        uint64_t* d; CK(hipMalloc(&d, h.size() * 8)); CK(hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice));
        a.mask_h = d; a.mask_v = d + (size_t)L * a.tiles * 64;
    }
    a.d_last = nullptr;          // as train_ops calls it (the heads are products of this kernel); a store here would sit in the DMA wait queue
    FusedBwdArgs fa{};
    fa.tape_h = dev_random<float>((size_t)L * n * 64, 6, 0.f, 1.f);
    fa.tape_feat = dev_random<float>((size_t)n * 64, 7, 0.f, 1.f);
    fa.enc_x = dev_random<float>((size_t)n * 64, 8, -1.f, 1.f);
    fa.enc_d = dev_random<float>((size_t)n * 64, 9, -1.f, 1.f);
    CK(hipMalloc(&fa.partial, (size_t)cus * FB_PART * 4));
    fa.skip_layer = sk;
    fa.w1t = dev_random<float>(64 * 64, 10, -0.1f, 0.1f);
    fa.dx = L == 4 ? 39 : 63;
    const int64_t iters = n / FB_ROWS;
    const int grid = (int)(iters < cus ? iters : cus);
    const double mfma_per_wave_iter = 32 + 64 + 64 * (L - 2) /* chain: layers_xyz[0]^T is applied once, in the epilogue */ + 32 + 32 + 64 + 64 * (L - 2) + (sk >= 0 ? 64 : 0) + 64 /* dW: the last delta has one product */ + 24 /* heads */;
    const double floor_us = mfma_per_wave_iter * 32 * 2 * (double)((iters + grid - 1) / grid) / 2.4e3;
    printf("L %d  n %lld  grid %d x 512  %d CUs;  %.0f MFMAs per wave and iteration: issue floor at 2.4 GHz %.1f us\n", L, (long long)n, grid, cus, mfma_per_wave_iter, floor_us);
#define RUN(M, A, what) printf("  %-58s %8.1f us\n", what, L <= 4 ? run<4, A>(a, fa, L, grid, 20) : run<8, A>(a, fa, L, grid, 20));
    RUN(4, 0, "the kernel as shipped")
    RUN(4, 2, "no activation-row DMA (stale LDS)")
    RUN(4, 1, "no dW products")
    RUN(4, 3, "no dW products, no row DMA (the delta chain + tile writes)")
    RUN(4, 4, "no chain MFMAs")
    RUN(4, 8, "no delta tile writes")
    RUN(4, 6, "no chain MFMAs, no row DMA (dW products from stale LDS)")
    RUN(4, 15, "barriers, waits, prologue and epilogue only")
    RUN(4, 6 + 16, "dW products alone, B operands not read")
    RUN(4, 6 + 32, "dW products alone, A operands not read")
    RUN(4, 6 + 48, "dW products alone, no operand reads (MFMAs + barriers)")
    RUN(4, 64, "everything, without the barrier between a delta's two phases")
    RUN(4, 2 + 128, "no row DMA, no waits, no barriers (the instruction streams alone)")
    RUN(4, 128, "row DMA issued but nothing waits for it: no waits, no barriers")
    RUN(4, 256, "row + weight DMA issued, barriers kept, no DMA waits at all")
    RUN(4, 256 + 2, "barriers kept, no row DMA, no DMA waits")
    RUN(4, 2 + 4 + 128, "dW products alone, no waits, no barriers")
    RUN(4, 2 + 4 + 48 + 128, "dW MFMAs alone: no operand reads, no waits, no barriers")
    RUN(4, 1 + 2 + 128, "the delta chain alone, no waits, no barriers")
    return 0;
}
