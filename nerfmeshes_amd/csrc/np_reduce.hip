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
// A full 8192-chunk is a perfect binary tree over 64 blocks of 128: one wavefront per chunk, lane = (block, half of the
// block's accumulators), shuffles for the trees.  The ragged last chunk runs the generic recursion on one lane (< 8192 elements).
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
        // lane = (block of this half, half of the block's 8 accumulators): accumulator j of a block sums the elements
        // 8 i + j in order of i -- four accumulators per lane, one 16-byte load per step (lane pairs read 32 contiguous
        // bytes, the 32 blocks of a half are 512 B apart; every byte of the chunk is fetched exactly once)
        float sub[2];
        float lo = INFINITY, hi = -INFINITY;
        bool nan = false;
        const int blk32 = lane >> 1, half = lane & 1;
#pragma unroll
        for (int it = 0; it < 2; ++it) {                // 32 blocks of 128 elements per iteration
            const float* p = a + base + (int64_t)(it * 32 + blk32) * 128 + 4 * half;
            float r[4];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                struct __attribute__((packed, aligned(4))) F4 { float x, y, z, w; };   // a rank's slice may start at any element
                const F4 v4 = *reinterpret_cast<const F4*>(p + 8 * i);
                const float v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (!SQDEV) { lo = fminf(lo, v[q]); hi = fmaxf(hi, v[q]); nan = nan || (v[q] != v[q]); }
                    float e = v[q];
                    if (SQDEV) { const float d = v[q] - mean; e = d * d; }
                    r[q] = i == 0 ? e : r[q] + e;
                }
            }
            // ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)): the two halves of a block, then the tree over the 32 blocks
            float t = (r[0] + r[1]) + (r[2] + r[3]);
            t += __shfl_xor(t, 1);
            t += __shfl_xor(t, 2);
            t += __shfl_xor(t, 4);
            t += __shfl_xor(t, 8);
            t += __shfl_xor(t, 16);
            t += __shfl_xor(t, 32);
            sub[it] = t;                                 // sum of elements [4096 it, 4096 (it + 1)) of the chunk
        }
        const float s = sub[0] + sub[1];
        if (lane == 0) csum[c] = s;
        if (!SQDEV) {
            for (int off = 32; off > 0; off >>= 1) { lo = fminf(lo, __shfl_xor(lo, off)); hi = fmaxf(hi, __shfl_xor(hi, off)); }
            if (__ballot(nan) != 0ull) lo = hi = NAN;    // numpy's minimum / maximum propagate NaN
            if (lane == 0) { cmin[c] = lo; cmax[c] = hi; }
        }
    }
}

// Sequential fp32 sum of 64 addends, one per lane, on top of `carry`: lane l ends up with the sum through addend l.
// One instruction per addend: P[l] = P[l - 1] + x[l] for ALL lanes at once (DPP wave_shr:1; lane 0, which has no source,
// keeps its value).  After step k lanes <= k hold their final value -- lane k because lane k - 1 did after step k - 1, the
// lanes in front because they recompute the same sum from inputs that no longer change -- and the lanes behind hold
// numbers nobody reads.  63 steps; the s_nop is the two wait states gfx9 wants between a VALU write and a DPP read of a VGPR.
#define NM_NP_STEP "v_add_f32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n"
#define NM_R3(x) x x x
#define NM_R4(x) x x x x
#define NM_R16(x) NM_R4(NM_R4(x))
__device__ __forceinline__ float np_chain64(float x, float carry, bool first, int lane) {
    float P = lane == 0 ? (first ? x : carry + x) : 0.0f;
    asm volatile("s_nop 1\n" NM_R3(NM_R16(NM_NP_STEP)) NM_R3(NM_R4(NM_NP_STEP)) NM_R3(NM_NP_STEP) : "+v"(P) : "v"(x));
    return P;
}

// the sequential fp32 accumulation of the chunk sums (what the reduction iterator does between buffer fills) + the
// final divisions; one workgroup, chunk sums staged through LDS in tiles
template <bool SQDEV>
__global__ __launch_bounds__(256) void np_finish_kernel(const float* __restrict__ csum, const float* __restrict__ cmin,
                                                        const float* __restrict__ cmax, int64_t chunks, int64_t n,
                                                        float* __restrict__ out /* [sum, mean, var, std, min, max] */) {
    // A chain of `chunks` dependent fp32 additions (13 500 at 480^3) on wavefront 0, every lane computing the same
    // accumulator.  The addends arrive 1024 at a time, one coalesced load per lane and 64 addends, the NEXT batch in
    // flight while the current one is added (a batch is ~2.5 us of chain, a load ~2 us for a lone wavefront), and reach the
    // chain one DPP addition each (np_chain64).  Measured per addend: lane 0 reading an LDS tile element by element 31
    // cycles; v_readlane + v_add_f32 with a scalar operand 21; the DPP form ~14 incl. the hand-over between batches.  Wavefronts 1-3 fold the
    // chunks' min / max (order-free) meanwhile.
    constexpr int U = 16;
    __shared__ float red[2][192];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave > 0) {
        if (!SQDEV) {
            float lo = INFINITY, hi = -INFINITY;
            const int t = threadIdx.x - 64;
            for (int64_t base = 0; base < chunks; base += 192 * 8) {
                float a[8], b[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int64_t i = base + 192 * u + t;
                    a[u] = i < chunks ? cmin[i] : INFINITY;
                    b[u] = i < chunks ? cmax[i] : -INFINITY;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) { lo = np_min(lo, a[u]); hi = np_max(hi, b[u]); }
            }
            red[0][t] = lo; red[1][t] = hi;
        }
    }
    float acc = 0.0f;
    if (wave == 0) {
        float cur[U], nxt[U];
        auto fetch = [&](float (&v)[U], int64_t base) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t i = base + 64 * u + lane;
                v[u] = i < chunks ? csum[i] : 0.0f;
            }
        };
        fetch(cur, 0);
        for (int64_t base = 0; base < chunks; base += 64 * U) {
            const int64_t left = chunks - base;
            if (left > 64 * U) fetch(nxt, base + 64 * U);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t m = left - 64 * u;                    // addends of this group of 64 (uniform)
                if (m > 0) {
                    const bool first = base == 0 && u == 0;
                    const float p63 = np_chain64(cur[u], acc, first, lane);
                    if (m >= 64) acc = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p63), 63));
                    else acc = __shfl(p63, (int)m - 1);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) cur[u] = nxt[u];
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (!SQDEV) {
            float lo = red[0][0], hi = red[1][0];
            for (int i = 1; i < 192; ++i) { lo = np_min(lo, red[0][i]); hi = np_max(hi, red[1][i]); }
            out[4] = lo; out[5] = hi;
        }
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
