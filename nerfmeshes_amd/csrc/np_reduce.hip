// numpy's fp32 reductions, bit for bit, on a device array: sum / mean / std as `ndarray.mean()` / `.std()` compute them.
//
// Why: the reference picks the marching-cubes level from numpy statistics of the density grid
// (/root/reference/src/mesh_nerf.py:56-65: iso = min(max(iso_level, min + std), max - std), all numpy fp32), and the
// mesh topology is a function of that level -- "bit-identical triangle topology for the same iso-level" needs the SAME
// level.  torch reductions (fp64 accumulation, different blocking) differ from numpy's in the last ulp, which is
// enough whenever the clamp is active (it is for the 480^3 bench grid: iso 18.096, not the requested 32).
//
// What numpy does (numpy/core/src/umath/loops_utils.h.src FLOAT_pairwise_sum + the reduction iterator; probed against
// numpy 2.2 in tests/test_np_reduce_restatement.py, unchanged since pairwise summation arrived in 1.9):
//   * the flat array is cut into buffer-sized chunks of 8192 elements; chunk sums are accumulated SEQUENTIALLY in fp32;
//   * inside a chunk: pairwise recursion, n -> (n/2 rounded down to a multiple of 8, rest), down to blocks of <= 128
//     elements; a block runs 8 interleaved accumulators r[j] += a[8 i + j] and combines them as
//     ((r0+r1)+(r2+r3)) + ((r4+r5)+(r6+r7)), then adds the < 8 left-over elements one by one; n < 8: plain loop;
//   * mean = sum / float32(n);  var: x = a - mean (fp32), x*x (fp32, two roundings), the same sum, / float32(n); sqrt.
// A full 8192-chunk is a perfect binary tree over 64 blocks of 128: one wavefront per chunk, lane = (block, accumulator),
// shuffles for the trees.  The ragged last chunk runs the generic recursion on one lane (< 8192 elements).
// HBM-bound: 2 passes over the array (8 B / element algorithmic).
#include "nm_internal.h"

namespace nm {

constexpr int NP_CHUNK = 8192;

// numpy's minimum / maximum reductions propagate NaN (fminf / fmaxf would drop it)
__device__ __forceinline__ float np_min(float a, float b) { return (a != a) ? a : ((b != b) ? b : fminf(a, b)); }
__device__ __forceinline__ float np_max(float a, float b) { return (a != a) ? a : ((b != b) ? b : fmaxf(a, b)); }

template <bool SQDEV>
__device__ __forceinline__ float np_elem(const float* __restrict__ a, int64_t i, float mean) {
    const float v = a[i];
    if (!SQDEV) return v;
    const float d = v - mean;
    return d * d;
}

template <bool SQDEV>
__device__ float np_block(const float* __restrict__ a, int64_t off, int n, float mean) {   // n <= 128
    if (n < 8) {
        float res = 0.0f;
        for (int i = 0; i < n; ++i) res += np_elem<SQDEV>(a, off + i, mean);
        return res;
    }
    float r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = np_elem<SQDEV>(a, off + j, mean);
    int i = 8;
    for (; i < n - (n % 8); i += 8)
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] += np_elem<SQDEV>(a, off + i + j, mean);
    float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += np_elem<SQDEV>(a, off + i, mean);
    return res;
}

template <bool SQDEV>
__device__ float np_pairwise(const float* __restrict__ a, int64_t off, int n, float mean) {   // generic, one thread
    struct Frame { int64_t off; int n; int stage; float left; };
    Frame st[24];
    int sp = 0;
    st[sp++] = Frame{off, n, 0, 0.0f};
    float ret = 0.0f;
    while (sp > 0) {
        Frame& f = st[sp - 1];
        if (f.n <= 128) { ret = np_block<SQDEV>(a, f.off, f.n, mean); --sp; continue; }
        int n2 = f.n / 2;
        n2 -= n2 % 8;
        if (f.stage == 0) { f.stage = 1; st[sp++] = Frame{f.off, n2, 0, 0.0f}; }
        else if (f.stage == 1) { f.left = ret; f.stage = 2; st[sp++] = Frame{f.off + n2, f.n - n2, 0, 0.0f}; }
        else { ret = f.left + ret; --sp; }
    }
    return ret;
}

// one wavefront per full chunk; also folds min / max of the raw values (exact, order-free) into the first pass
template <bool SQDEV>
__global__ __launch_bounds__(64) void np_chunk_sums_kernel(const float* __restrict__ a, int64_t full_chunks, int64_t n,
                                                           int64_t chunk_lo, int64_t chunk_hi,
                                                           const float* __restrict__ mean_ptr, float* __restrict__ csum,
                                                           float* __restrict__ cmin, float* __restrict__ cmax) {
    // `a` is addressed with GLOBAL element indices (the caller offsets the pointer when it holds a slice of the array);
    // chunks [chunk_lo, chunk_hi) of the global array are summed, results at csum[chunk - chunk_lo]
    const int lane = threadIdx.x;
    const int blk = lane >> 3, j = lane & 7;          // block of this iteration's 8, accumulator
    const float mean = SQDEV ? *mean_ptr : 0.0f;
    csum -= chunk_lo; cmin -= chunk_lo; cmax -= chunk_lo;
    for (int64_t c = chunk_lo + blockIdx.x; c < chunk_hi; c += gridDim.x) {
        const int64_t base = c * NP_CHUNK;
        if (c >= full_chunks) {                         // ragged tail: generic recursion on one lane
            const int m = (int)(n - base);
            if (lane == 0) csum[c] = np_pairwise<SQDEV>(a, base, m, mean);
            if (!SQDEV) {
                float lo = INFINITY, hi = -INFINITY;
                for (int i = lane; i < m; i += 64) { const float v = a[base + i]; lo = np_min(lo, v); hi = np_max(hi, v); }
                for (int off = 32; off > 0; off >>= 1) { lo = np_min(lo, __shfl_xor(lo, off)); hi = np_max(hi, __shfl_xor(hi, off)); }
                if (lane == 0) { cmin[c] = lo; cmax[c] = hi; }
            }
            continue;
        }
        float sub[8];
        float lo = INFINITY, hi = -INFINITY;
#pragma unroll
        for (int it = 0; it < 8; ++it) {                // 8 blocks of 128 elements per iteration
            const float* p = a + base + (int64_t)(it * 8 + blk) * 128 + j;
            float r = 0.0f;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float v = p[8 * i];
                if (!SQDEV) { lo = np_min(lo, v); hi = np_max(hi, v); }
                float e = v;
                if (SQDEV) { const float d = v - mean; e = d * d; }
                r = i == 0 ? e : r + e;
            }
            // ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) over the 8 accumulator lanes, then the tree over the 8 blocks
            r += __shfl_xor(r, 1);
            r += __shfl_xor(r, 2);
            r += __shfl_xor(r, 4);
            r += __shfl_xor(r, 8);
            r += __shfl_xor(r, 16);
            r += __shfl_xor(r, 32);
            sub[it] = r;                                 // sum of elements [1024 it, 1024 (it + 1)) of the chunk
        }
        const float s = ((sub[0] + sub[1]) + (sub[2] + sub[3])) + ((sub[4] + sub[5]) + (sub[6] + sub[7]));
        if (lane == 0) csum[c] = s;
        if (!SQDEV) {
            for (int off = 32; off > 0; off >>= 1) { lo = np_min(lo, __shfl_xor(lo, off)); hi = np_max(hi, __shfl_xor(hi, off)); }
            if (lane == 0) { cmin[c] = lo; cmax[c] = hi; }
        }
    }
}

// the sequential fp32 accumulation of the chunk sums (what the reduction iterator does between buffer fills) + the
// final divisions; one workgroup, chunk sums staged through LDS in tiles
template <bool SQDEV>
__global__ __launch_bounds__(256) void np_finish_kernel(const float* __restrict__ csum, const float* __restrict__ cmin,
                                                        const float* __restrict__ cmax, int64_t chunks, int64_t n,
                                                        float* __restrict__ out /* [sum, mean, var, std, min, max] */) {
    __shared__ float tile[8192];
    __shared__ float red[2][256];
    float acc = 0.0f;
    float lo = INFINITY, hi = -INFINITY;
    for (int64_t base = 0; base < chunks; base += 8192) {
        const int m = (int)((chunks - base) < 8192 ? (chunks - base) : 8192);
        for (int i = threadIdx.x; i < m; i += 256) {
            tile[i] = csum[base + i];
            if (!SQDEV) { lo = np_min(lo, cmin[base + i]); hi = np_max(hi, cmax[base + i]); }
        }
        __syncthreads();
        if (threadIdx.x == 0)
            for (int i = 0; i < m; ++i) acc = (base == 0 && i == 0) ? tile[0] : acc + tile[i];
        __syncthreads();
    }
    if (!SQDEV) {
        red[0][threadIdx.x] = lo; red[1][threadIdx.x] = hi;
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int i = 1; i < 256; ++i) { lo = np_min(lo, red[0][i]); hi = np_max(hi, red[1][i]); }
            out[4] = lo; out[5] = hi;
        }
    }
    if (threadIdx.x == 0) {
        const float count = (float)n;                   // float32(n), round to nearest, as numpy casts the divisor
        if (!SQDEV) { out[0] = acc; out[1] = acc / count; }
        else { const float var = acc / count; out[2] = var; out[3] = sqrtf(var); }
    }
}

}  // namespace nm

using namespace nm;

extern "C" int64_t nm_np_stats_workspace_bytes(int64_t n) {
    if (n <= 0) return 0;
    const int64_t chunks = (n + NP_CHUNK - 1) / NP_CHUNK;
    return (3 * chunks + 16) * 4;
}

extern "C" int nm_np_stats(const float* d_x, int64_t n, void* d_workspace, float* h_out6, void* stream_) {
    NM_REQUIRE(d_x && d_workspace && h_out6 && n > 0, "bad argument");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int64_t full = n / NP_CHUNK, chunks = (n + NP_CHUNK - 1) / NP_CHUNK;
    float* csum = static_cast<float*>(d_workspace);
    float* cmin = csum + chunks;
    float* cmax = cmin + chunks;
    float* out = cmax + chunks;                          // 6 floats (+ padding)
    const unsigned grid = (unsigned)(chunks < 65536 ? chunks : 65536);
    hipLaunchKernelGGL(np_chunk_sums_kernel<false>, dim3(grid), dim3(64), 0, stream, d_x, full, n, (int64_t)0, chunks,
                       (const float*)nullptr, csum, cmin, cmax);
    hipLaunchKernelGGL(np_finish_kernel<false>, dim3(1), dim3(256), 0, stream, csum, cmin, cmax, chunks, n, out);
    hipLaunchKernelGGL(np_chunk_sums_kernel<true>, dim3(grid), dim3(64), 0, stream, d_x, full, n, (int64_t)0, chunks,
                       (const float*)(out + 1), csum, cmin, cmax);
    hipLaunchKernelGGL(np_finish_kernel<true>, dim3(1), dim3(256), 0, stream, csum, cmin, cmax, chunks, n, out);
    NM_HIP_CHECK(hipGetLastError());
    NM_HIP_CHECK(hipMemcpyAsync(h_out6, out, 6 * sizeof(float), hipMemcpyDeviceToHost, stream));
    NM_HIP_CHECK(hipStreamSynchronize(stream));
    return 0;
}

// The same statistics of an array that is spread over several devices (axis-0 slabs of the density grid): the two stages
// separately.  A rank holds the global elements [first, first + count) and sums the 8192-element chunks
// [chunk_lo, chunk_hi) of the GLOBAL array (every element of those chunks must be among the ones it holds);
// nm_np_chunk_count(n) chunks exist.  The chunk sums of all ranks, concatenated in chunk order, go through nm_np_finish --
// the sequential fp32 accumulation numpy's reduction iterator performs -- on every rank alike.
extern "C" int64_t nm_np_chunk_count(int64_t n) { return n > 0 ? (n + NP_CHUNK - 1) / NP_CHUNK : 0; }

extern "C" int nm_np_chunk_sums(const float* d_x, int64_t first, int64_t count, int64_t n, int64_t chunk_lo, int64_t chunk_hi,
                                int32_t squared_deviation, float mean, float* d_csum, float* d_cmin, float* d_cmax,
                                void* stream_) {
    NM_REQUIRE(d_x && d_csum && n > 0 && count > 0 && first >= 0 && first + count <= n, "bad argument");
    NM_REQUIRE(squared_deviation || (d_cmin && d_cmax), "np_chunk_sums: the first pass also returns the chunks' min / max");
    const int64_t chunks = (n + NP_CHUNK - 1) / NP_CHUNK;
    NM_REQUIRE(chunk_lo >= 0 && chunk_lo <= chunk_hi && chunk_hi <= chunks, "np_chunk_sums: bad chunk range");
    if (chunk_hi == chunk_lo) return 0;
    const int64_t need_hi = chunk_hi * NP_CHUNK < n ? chunk_hi * NP_CHUNK : n;
    NM_REQUIRE(chunk_lo * NP_CHUNK >= first && need_hi <= first + count, "np_chunk_sums: the chunks reach outside the held elements");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int64_t m = chunk_hi - chunk_lo;
    const unsigned grid = (unsigned)(m < 65536 ? m : 65536);
    float* d_mean = nullptr;
    if (squared_deviation) {            // the kernel reads the mean from device memory: park it behind the caller's sums
        d_mean = d_csum + m;            // (the caller allocates m + 1 floats for the second pass)
        NM_HIP_CHECK(hipMemcpyAsync(d_mean, &mean, sizeof(float), hipMemcpyHostToDevice, stream));
        hipLaunchKernelGGL(np_chunk_sums_kernel<true>, dim3(grid), dim3(64), 0, stream, d_x - first, n / NP_CHUNK, n, chunk_lo,
                           chunk_hi, (const float*)d_mean, d_csum, d_cmin, d_cmax);
        NM_HIP_CHECK(hipStreamSynchronize(stream));      // `mean` lives on the caller's stack
    } else {
        hipLaunchKernelGGL(np_chunk_sums_kernel<false>, dim3(grid), dim3(64), 0, stream, d_x - first, n / NP_CHUNK, n, chunk_lo,
                           chunk_hi, (const float*)nullptr, d_csum, d_cmin, d_cmax);
    }
    NM_HIP_CHECK(hipGetLastError());
    return 0;
}

extern "C" int nm_np_finish(const float* d_csum, const float* d_cmin, const float* d_cmax, int64_t chunks, int64_t n,
                            int32_t squared_deviation, float* d_out6, float* h_out6, void* stream_) {
    NM_REQUIRE(d_csum && d_out6 && h_out6 && chunks > 0 && n > 0, "bad argument");
    NM_REQUIRE(squared_deviation || (d_cmin && d_cmax), "np_finish: the first pass folds the chunks' min / max");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (squared_deviation)
        hipLaunchKernelGGL(np_finish_kernel<true>, dim3(1), dim3(256), 0, stream, d_csum, d_cmin, d_cmax, chunks, n, d_out6);
    else
        hipLaunchKernelGGL(np_finish_kernel<false>, dim3(1), dim3(256), 0, stream, d_csum, d_cmin, d_cmax, chunks, n, d_out6);
    NM_HIP_CHECK(hipGetLastError());
    NM_HIP_CHECK(hipMemcpyAsync(h_out6, d_out6, 6 * sizeof(float), hipMemcpyDeviceToHost, stream));
    NM_HIP_CHECK(hipStreamSynchronize(stream));
    return 0;
}
