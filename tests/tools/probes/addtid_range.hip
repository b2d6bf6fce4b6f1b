// Probe (MI355X): range checking of buffer_load ... lds through an ADD_TID_ENABLE descriptor (stride = 16, no VGPR address).
// Which lanes come back zero for a given NUM_RECORDS and soffset?  hipcc --offload-arch=gfx950 addtid_range.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(const float* src, float* out, int num_records, int soffset, int stride) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    for (int i = threadIdx.x; i < 512; i += 64) reinterpret_cast<float*>(lds)[i] = -7.0f;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), (short)stride, num_records, 1 << 23);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds, 16, 0, soffset, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) out[i] = reinterpret_cast<float*>(lds)[i];
}
int main() {
    std::vector<float> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = 1.0f + i;
    float *d, *o;
    hipMalloc(&d, 4096 * 4); hipMalloc(&o, 256 * 4);
    hipMemcpy(d, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    const int cases[][2] = {{0, 0}, {1, 0}, {5, 0}, {16, 0}, {63, 0}, {64, 0}, {80, 0}, {160, 0}, {1024, 0}, {5, 256}, {16, 256}, {21, 256}, {64, 256},
                            {80, 256}, {1 << 20, 256}, {336, 256}, {1024 + 80, 1024}};
    for (auto& c : cases) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 4096, 0, d, o, c[0], c[1], 16);
        std::vector<float> r(256);
        hipMemcpy(r.data(), o, 256 * 4, hipMemcpyDeviceToHost);
        int valid = 0, zero = 0, first_zero = -1;
        for (int l = 0; l < 64; ++l) {
            const bool ok = r[4 * l] == 1.0f + c[1] / 4 + 4 * l;
            valid += ok; zero += r[4 * l] == 0.0f;
            if (!ok && first_zero < 0) first_zero = l;
        }
        printf("num_records %8d soffset %5d: lanes valid %2d zero %2d first non-valid lane %d (lane0 value %g)\n", c[0], c[1], valid, zero, first_zero, r[0]);
    }
    return 0;
}
