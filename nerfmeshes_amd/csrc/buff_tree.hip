// BuFF voxel-tree sampler for gfx950: TreeSampling.batch_ray_voxel_intersect, deterministic branch
// (/root/reference/src/nerf/tree.py:215-343) as ONE kernel, one 64-lane wavefront per ray.
//
// The reference materialises dense (R, N, 3) slab-test tensors (2 x 42 MB at R = 2048, N = 1728), sorts all
// N voxels per ray three times and gathers five times.  Here a wave streams the N axis-aligned boxes (41 KB,
// L2 resident) through the slab test, compacts the handful of boxes the ray actually crosses into LDS with a
// ballot, orders them by entry depth, and places the S samples:
//   samples = linspace(0,1,S) * (total crossed length); bucket = first box whose cumulative length reaches
//   the sample; z = t_enter(bucket) + (sample - first sample of the same bucket); final sort by z.
// z values and the ray mask are bit-identical to the reference (same fp32 expressions, cumulative lengths
// accumulated in fp64 and rounded per element like torch.cumsum).  Voxel ids follow a STABLE ordering of
// ties; the reference leaves that order to torch.sort's unstable default, which makes its ids inconsistent
// with its own z values on ~90 % of samples (tests/test_oracle_golden.py::test_buff_...).
#include "nm_internal.h"

namespace nm {

constexpr int BUFF_MAX_HITS = 512;     // boxes one ray can cross (12^3 root grid: <= 34; refined trees more)
constexpr int BUFF_MAX_SAMPLES = 512;

__device__ __forceinline__ double shfl_up_d(double v, int delta) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_up(lo, delta); hi = __shfl_up(hi, delta);
    return __hiloint2double(hi, lo);
}

__global__ __launch_bounds__(256) void buff_intersect_kernel(const float* __restrict__ voxels, int nvox,
                                                             const float* __restrict__ origins, int origins_per_ray,
                                                             const float* __restrict__ dirs, float near_, float far_,
                                                             const float* __restrict__ u, int64_t rays, int samples,
                                                             float* __restrict__ z_out, int64_t* __restrict__ idx_out,
                                                             uint8_t* __restrict__ mask_out, int* __restrict__ overflow) {
    __shared__ float h_tmin[4][BUFF_MAX_HITS], h_tmax[4][BUFF_MAX_HITS], s_tmin[4][BUFF_MAX_HITS], s_cum[4][BUFF_MAX_HITS];
    __shared__ int h_id[4][BUFF_MAX_HITS], s_id[4][BUFF_MAX_HITS];
    __shared__ float p_s[4][BUFF_MAX_SAMPLES], p_z[4][BUFF_MAX_SAMPLES];
    __shared__ int p_bucket[4][BUFF_MAX_SAMPLES], p_vid[4][BUFF_MAX_SAMPLES];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int64_t ray = (int64_t)blockIdx.x * 4 + wv; ray < rays; ray += (int64_t)gridDim.x * 4) {
        const float* o = origins + (origins_per_ray ? 3 * ray : 0);
        float inv[3], org[3];
        int sgn[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            org[a] = o[a];
            inv[a] = 1.0f / dirs[3 * ray + a];
            sgn[a] = inv[a] < 0.0f ? 1 : 0;
        }
        // ---- slab test over all boxes, ballot-compaction of the crossed ones (in box-index order)
        int K = 0;
        float best_t = 0.0f;
        int best_i = 0x7fffffff;
        bool have_best = false;
        for (int base = 0; base < nvox; base += 64) {
            const int n = base + lane;
            bool valid = false;
            float tmin = 0.0f, tmax = 0.0f;
            if (n < nvox) {
                const float* b = voxels + 6 * (int64_t)n;   // [min xyz | max xyz]
                float lo_t[3], hi_t[3];
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    lo_t[a] = (b[3 * sgn[a] + a] - org[a]) * inv[a];
                    hi_t[a] = (b[3 * (1 - sgn[a]) + a] - org[a]) * inv[a];
                }
                valid = (lo_t[0] <= hi_t[1]) && (lo_t[1] <= hi_t[0]);
                tmin = lo_t[1] > lo_t[0] ? lo_t[1] : lo_t[0];
                tmax = hi_t[1] < hi_t[0] ? hi_t[1] : hi_t[0];
                valid = valid && (tmin <= hi_t[2]) && (lo_t[2] <= tmax);
                tmin = lo_t[2] > tmin ? lo_t[2] : tmin;
                tmax = hi_t[2] < tmax ? hi_t[2] : tmax;
                valid = valid && (tmin >= near_) && (tmax <= far_);
                // stable argmin of tmin over ALL boxes: what the reference reports for rays that miss everything
                if (!have_best || tmin < best_t) { best_t = tmin; best_i = n; have_best = true; }
            }
            const unsigned long long bal = __ballot(valid);
            if (valid) {
                const int pos = K + __popcll(bal & ((1ull << lane) - 1ull));
                if (pos < BUFF_MAX_HITS) { h_tmin[wv][pos] = tmin; h_tmax[wv][pos] = tmax; h_id[wv][pos] = n; }
            }
            K += __popcll(bal);
        }
        if (K > BUFF_MAX_HITS) { if (lane == 0) atomicExch(overflow, 1); K = BUFF_MAX_HITS; }
        // wave argmin (ties -> lowest index); NaN entry depths never win, as in a stable ascending sort
        for (int off = 32; off > 0; off >>= 1) {
            const float ot = __shfl_xor(best_t, off);
            const int oi = __shfl_xor(best_i, off);
            const bool oh = __shfl_xor((int)have_best, off) != 0;
            if (oh && (!have_best || ot < best_t || (ot == best_t && oi < best_i))) { best_t = ot; best_i = oi; have_best = true; }
        }
        __builtin_amdgcn_wave_barrier();
        // ---- order the crossed boxes by entry depth (rank sort, ties keep box-index order)
        for (int i = lane; i < K; i += 64) {
            const float t = h_tmin[wv][i];
            int rank = 0;
            for (int k = 0; k < K; ++k) {
                const float ot = h_tmin[wv][k];
                rank += (ot < t || (ot == t && k < i)) ? 1 : 0;
            }
            s_tmin[wv][rank] = t;
            s_cum[wv][rank] = h_tmax[wv][i] - t;     // crossed length, turned into its running sum below
            s_id[wv][rank] = h_id[wv][i];
        }
        __builtin_amdgcn_wave_barrier();
        // ---- cumulative crossed length: fp64 accumulation, fp32 per element (torch.cumsum on CPU)
        double carry = 0.0;
        for (int base = 0; base < K; base += 64) {
            const int i = base + lane;
            double incl = i < K ? (double)s_cum[wv][i] : 0.0;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const double p = shfl_up_d(incl, off);
                if (lane >= off) incl += p;
            }
            incl += carry;
            if (i < K) s_cum[wv][i] = (float)incl;
            const int lo = __double2loint(incl), hi = __double2hiint(incl);
            carry = __hiloint2double(__shfl(hi, 63), __shfl(lo, 63));
        }
        __builtin_amdgcn_wave_barrier();
        const float total = K > 0 ? s_cum[wv][K - 1] : 0.0f;
        // ---- place the samples
        for (int j = lane; j < samples; j += 64) {
            const float s = u[j] * total;
            int lo = 0, hi = K > 0 ? K - 1 : 0;     // first i with cum[i] >= s (cum[K-1] == total >= s)
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (s_cum[wv][mid] < s) lo = mid + 1; else hi = mid;
            }
            p_s[wv][j] = s;
            p_bucket[wv][j] = lo;
        }
        __builtin_amdgcn_wave_barrier();
        for (int j = lane; j < samples; j += 64) {
            const int bkt = p_bucket[wv][j];
            int lo = 0, hi = j;                      // first sample of the same bucket
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (p_bucket[wv][mid] < bkt) lo = mid + 1; else hi = mid;
            }
            const float offset = p_s[wv][j] - p_s[wv][lo];
            p_z[wv][j] = (K > 0 ? s_tmin[wv][bkt] : 0.0f) + offset;
            p_vid[wv][j] = K > 0 ? s_id[wv][bkt] : best_i;
        }
        __builtin_amdgcn_wave_barrier();
        // ---- final ascending sort of the depths (stable), voxel ids follow
        for (int j = lane; j < samples; j += 64) {
            const float zj = p_z[wv][j];
            int rank = 0;
            for (int k = 0; k < samples; ++k) {
                const float zk = p_z[wv][k];
                rank += (zk < zj || (zk == zj && k < j)) ? 1 : 0;
            }
            z_out[ray * samples + rank] = zj;
            idx_out[ray * samples + rank] = p_vid[wv][j];
        }
        if (lane == 0) mask_out[ray] = K > 0 ? 1 : 0;
        __builtin_amdgcn_wave_barrier();
    }
}


// ---- `tree.use_random_sampling` branch (tree.py:280-297, :337-341) as a function of the draws -------------------------
// The reference gives crossed voxels weight 1 and all others 1e-12, draws S voxels per ray with
// torch.multinomial(weights, S, replacement=True) and a depth uniformly inside each drawn voxel's [t_enter, t_exit].
// torch.multinomial with replacement is an inverse-CDF sampler (ATen/native/cpu/MultinomialKernel.cpp): the fp32
// running sum of the row divided by its total, one double u per sample, lower-bound search for the first category
// whose cumulative probability is >= u.  The running sum is the number of crossed voxels so far (1e-12 is absorbed
// once the sum reaches 1), so the draw is the c-th crossed voxel in index order with c the smallest count whose
// fp32 quotient c / K is >= u.  The draws are the caller's (u_pick (R,S) double, u_pos (R,S) float: torch's
// generator), the arithmetic is the reference's: given the reference's own draws the depths and voxel ids are its
// output bit for bit (tests/golden/buff_random.npz) -- except for a draw below the cumulative share of the 1e-12
// weights in front of the first crossed voxel (u < 1.7e-9), where the reference returns a voxel the ray does not cross
// and this kernel the first crossed one.  Rows of rays that cross nothing are zero-filled: the
// reference samples arbitrary voxels there, BuFFModel.forward overwrites those depths and nothing reads the ids.
__global__ __launch_bounds__(256) void buff_random_kernel(const float* __restrict__ voxels, int nvox,
                                                          const float* __restrict__ origins, int origins_per_ray,
                                                          const float* __restrict__ dirs, float near_, float far_,
                                                          const double* __restrict__ u_pick, const float* __restrict__ u_pos,
                                                          int64_t rays, int samples, float* __restrict__ z_out,
                                                          int64_t* __restrict__ idx_out, uint8_t* __restrict__ mask_out,
                                                          int* __restrict__ overflow) {
    __shared__ float h_tmin[4][BUFF_MAX_HITS], h_tmax[4][BUFF_MAX_HITS];
    __shared__ int h_id[4][BUFF_MAX_HITS];
    __shared__ float p_z[4][BUFF_MAX_SAMPLES];
    __shared__ int p_vid[4][BUFF_MAX_SAMPLES];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int64_t ray = (int64_t)blockIdx.x * 4 + wv; ray < rays; ray += (int64_t)gridDim.x * 4) {
        const float* o = origins + (origins_per_ray ? 3 * ray : 0);
        float inv[3], org[3];
        int sgn[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            org[a] = o[a];
            inv[a] = 1.0f / dirs[3 * ray + a];
            sgn[a] = inv[a] < 0.0f ? 1 : 0;
        }
        // ---- slab test over all boxes (the expressions of buff_intersect_kernel), crossed ones compacted in index order
        int K = 0;
        for (int base = 0; base < nvox; base += 64) {
            const int n = base + lane;
            bool valid = false;
            float tmin = 0.0f, tmax = 0.0f;
            if (n < nvox) {
                const float* b = voxels + 6 * (int64_t)n;
                float lo_t[3], hi_t[3];
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    lo_t[a] = (b[3 * sgn[a] + a] - org[a]) * inv[a];
                    hi_t[a] = (b[3 * (1 - sgn[a]) + a] - org[a]) * inv[a];
                }
                valid = (lo_t[0] <= hi_t[1]) && (lo_t[1] <= hi_t[0]);
                tmin = lo_t[1] > lo_t[0] ? lo_t[1] : lo_t[0];
                tmax = hi_t[1] < hi_t[0] ? hi_t[1] : hi_t[0];
                valid = valid && (tmin <= hi_t[2]) && (lo_t[2] <= tmax);
                tmin = lo_t[2] > tmin ? lo_t[2] : tmin;
                tmax = hi_t[2] < tmax ? hi_t[2] : tmax;
                valid = valid && (tmin >= near_) && (tmax <= far_);
            }
            const unsigned long long bal = __ballot(valid);
            if (valid) {
                const int pos = K + __popcll(bal & ((1ull << lane) - 1ull));
                if (pos < BUFF_MAX_HITS) { h_tmin[wv][pos] = tmin; h_tmax[wv][pos] = tmax; h_id[wv][pos] = n; }
            }
            K += __popcll(bal);
        }
        if (K > BUFF_MAX_HITS) { if (lane == 0) atomicExch(overflow, 1); K = BUFF_MAX_HITS; }
        __builtin_amdgcn_wave_barrier();
        // ---- draw: voxel by inverse CDF over the crossed ones, depth uniformly inside it
        for (int j = lane; j < samples; j += 64) {
            float z = 0.0f;
            int vid = 0;
            if (K > 0) {
                const double u = u_pick[ray * samples + j];
                const float fk = (float)K;
                int lo = 1, hi = K;                   // smallest count c with fp32(c / K) >= u; fp32(K / K) = 1 > u
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if ((double)((float)mid / fk) < u) lo = mid + 1; else hi = mid;
                }
                const float a = h_tmin[wv][lo - 1], b = h_tmax[wv][lo - 1];
                z = a + (b - a) * u_pos[ray * samples + j];       // two roundings after the subtraction, as torch
                vid = h_id[wv][lo - 1];
            }
            p_z[wv][j] = z;
            p_vid[wv][j] = vid;
        }
        __builtin_amdgcn_wave_barrier();
        // ---- ascending sort of the depths (stable; the reference's unstable sort differs only on exact ties)
        for (int j = lane; j < samples; j += 64) {
            const float zj = p_z[wv][j];
            int rank = 0;
            for (int k = 0; k < samples; ++k) {
                const float zk = p_z[wv][k];
                rank += (zk < zj || (zk == zj && k < j)) ? 1 : 0;
            }
            z_out[ray * samples + rank] = zj;
            idx_out[ray * samples + rank] = p_vid[wv][j];
        }
        if (lane == 0) mask_out[ray] = K > 0 ? 1 : 0;
        __builtin_amdgcn_wave_barrier();
    }
}


// ---- opt-in: voxel ids with the reference's OWN tie order ---------------------------------------------------------
// The reference calls torch.sort three times with its unstable default (tree.py:300, :306, :335).  On the CPU build it
// was written against (and the golden vectors were generated with: torch 2.10, tests/golden/make_golden.py) that is
// libstdc++'s std::sort -- introsort -- over (key, index) pairs
// (ATen/native/cpu/SortingKernel.cpp: std::sort(composite accessor, KeyValueCompAsc / KeyValueCompDesc)), a
// deterministic algorithm: probed equal for ATEN_CPU_CAPABILITY = default / avx2 / avx512, and a line-by-line
// restatement reproduces torch.sort's indices on tie-heavy inputs (tests/test_introsort_restatement.py).  The order of
// ties decides (a) the order of equal entry depths -- common: every voxel of a slab shares the slab's entry plane --
// hence the 0/1 hit sequence, (b) which crossed voxel each slot of the "rolled to the front" list is attributed to
// (tree.py:306-309: the VALUES are placed by a boolean mask, in order; the INDICES come from the unstable sort), and
// (c) the order of samples with equal depth.  This kernel replays exactly that: one wavefront per ray, the three
// introsorts run on lane 0 over LDS (they are sequential algorithms), everything else is wave-parallel.  It exists
// for parity (ids == the reference's, bit for bit, also on rays where they are not the voxels the samples lie in);
// the default kernel above keeps the stable, geometrically meaningful order and is ~100x faster.
struct CmpAsc {   // KeyValueCompAsc<float>: NaNs last
    __device__ __forceinline__ bool operator()(float a, float b) const { return (!(a != a) && (b != b)) || (a < b); }
};
struct CmpDesc {  // KeyValueCompDesc<float>: NaNs first
    __device__ __forceinline__ bool operator()(float a, float b) const { return ((a != a) && !(b != b)) || (a > b); }
};

template <typename Cmp>
struct Introsort {   // libstdc++ bits/stl_algo.h: __sort = __introsort_loop + __final_insertion_sort, threshold 16
    float* k;
    unsigned short* ix;
    Cmp cmp;
    __device__ __forceinline__ void swap(int i, int j) {
        const float tk = k[i]; k[i] = k[j]; k[j] = tk;
        const unsigned short ti = ix[i]; ix[i] = ix[j]; ix[j] = ti;
    }
    __device__ void move_median_to_first(int result, int a, int b, int c) {
        if (cmp(k[a], k[b])) {
            if (cmp(k[b], k[c])) swap(result, b);
            else if (cmp(k[a], k[c])) swap(result, c);
            else swap(result, a);
        } else if (cmp(k[a], k[c])) swap(result, a);
        else if (cmp(k[b], k[c])) swap(result, c);
        else swap(result, b);
    }
    __device__ int unguarded_partition(int first, int last, int pivot) {
        for (;;) {
            while (cmp(k[first], k[pivot])) ++first;
            --last;
            while (cmp(k[pivot], k[last])) --last;
            if (!(first < last)) return first;
            swap(first, last);
            ++first;
        }
    }
    // heap fallback (__partial_sort(first, last, last) = make_heap + sort_heap) when the depth limit is reached
    __device__ void push_heap(int first, int hole, int top, float vk, unsigned short vi) {
        int parent = (hole - 1) / 2;
        while (hole > top && cmp(k[first + parent], vk)) {
            k[first + hole] = k[first + parent]; ix[first + hole] = ix[first + parent];
            hole = parent; parent = (hole - 1) / 2;
        }
        k[first + hole] = vk; ix[first + hole] = vi;
    }
    __device__ void adjust_heap(int first, int hole, int len, float vk, unsigned short vi) {
        const int top = hole;
        int child = hole;
        while (child < (len - 1) / 2) {
            child = 2 * (child + 1);
            if (cmp(k[first + child], k[first + child - 1])) --child;
            k[first + hole] = k[first + child]; ix[first + hole] = ix[first + child];
            hole = child;
        }
        if ((len & 1) == 0 && child == (len - 2) / 2) {
            child = 2 * (child + 1);
            k[first + hole] = k[first + child - 1]; ix[first + hole] = ix[first + child - 1];
            hole = child - 1;
        }
        push_heap(first, hole, top, vk, vi);
    }
    __device__ void heap_sort(int first, int last) {
        const int len = last - first;
        if (len >= 2) {
            for (int parent = (len - 2) / 2;; --parent) {
                adjust_heap(first, parent, len, k[first + parent], ix[first + parent]);
                if (parent == 0) break;
            }
        }
        while (last - first > 1) {
            --last;
            const float vk = k[last]; const unsigned short vi = ix[last];
            k[last] = k[first]; ix[last] = ix[first];
            adjust_heap(first, 0, last - first, vk, vi);
        }
    }
    __device__ void unguarded_linear_insert(int last) {
        const float vk = k[last]; const unsigned short vi = ix[last];
        int next = last - 1;
        while (cmp(vk, k[next])) { k[last] = k[next]; ix[last] = ix[next]; last = next; --next; }
        k[last] = vk; ix[last] = vi;
    }
    __device__ void insertion_sort(int first, int last) {
        if (first == last) return;
        for (int i = first + 1; i != last; ++i) {
            if (cmp(k[i], k[first])) {
                const float vk = k[i]; const unsigned short vi = ix[i];
                for (int j = i; j > first; --j) { k[j] = k[j - 1]; ix[j] = ix[j - 1]; }
                k[first] = vk; ix[first] = vi;
            } else unguarded_linear_insert(i);
        }
    }
    __device__ void sort(int n) {
        if (n <= 0) return;
        int lg = 0;
        while ((2 << lg) <= n) ++lg;          // std::__lg(n)
        // __introsort_loop: recursion on the right part, iteration on the left; disjoint ranges, so an explicit stack
        // in any order gives the same result
        int st_first[64], st_last[64], st_depth[64], sp = 0;
        st_first[0] = 0; st_last[0] = n; st_depth[0] = 2 * lg; sp = 1;
        while (sp > 0) {
            --sp;
            int first = st_first[sp], last = st_last[sp], depth = st_depth[sp];
            while (last - first > 16) {
                if (depth == 0) { heap_sort(first, last); break; }
                --depth;
                const int mid = first + (last - first) / 2;
                move_median_to_first(first, first + 1, mid, last - 1);
                const int cut = unguarded_partition(first + 1, last, first);
                st_first[sp] = cut; st_last[sp] = last; st_depth[sp] = depth; ++sp;
                last = cut;
            }
        }
        if (n > 16) {
            insertion_sort(0, 16);
            for (int i = 16; i < n; ++i) unguarded_linear_insert(i);
        } else insertion_sort(0, n);
    }
};

__global__ __launch_bounds__(64) void buff_reference_ids_kernel(const float* __restrict__ voxels, int nvox, int npad,
                                                               const float* __restrict__ origins, int origins_per_ray,
                                                               const float* __restrict__ dirs, float near_, float far_,
                                                               const float* __restrict__ u, int64_t rays, int samples,
                                                               float* __restrict__ z_out, int64_t* __restrict__ idx_out,
                                                               uint8_t* __restrict__ mask_out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* key = reinterpret_cast<float*>(smem);                       // [npad] sort keys
    float* r_tmin = key + npad;                                        // [npad] crossed boxes in reference order
    float* r_cum = r_tmin + npad;                                      // [npad]
    float* p_s = r_cum + npad;                                         // [BUFF_MAX_SAMPLES]
    float* p_z = p_s + BUFF_MAX_SAMPLES;
    int* p_bucket = reinterpret_cast<int*>(p_z + BUFF_MAX_SAMPLES);
    unsigned short* perm1 = reinterpret_cast<unsigned short*>(p_bucket + BUFF_MAX_SAMPLES);   // crosses_sorted.indices
    unsigned short* perm2 = perm1 + npad;                                                     // crosses_start.indices
    unsigned short* p_ix = perm2 + npad;                               // [BUFF_MAX_SAMPLES] z sort indices
    unsigned short* p_vid = p_ix + BUFF_MAX_SAMPLES;
    unsigned char* hit = reinterpret_cast<unsigned char*>(p_vid + BUFF_MAX_SAMPLES);          // [npad] by box index
    const int lane = threadIdx.x;
    for (int64_t ray = blockIdx.x; ray < rays; ray += gridDim.x) {
        const float* o = origins + (origins_per_ray ? 3 * ray : 0);
        float inv[3], org[3];
        int sgn[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            org[a] = o[a];
            inv[a] = 1.0f / dirs[3 * ray + a];
            sgn[a] = inv[a] < 0.0f ? 1 : 0;
        }
        auto slab = [&](int n, float& tmin, float& tmax) -> bool {
            const float* b = voxels + 6 * (int64_t)n;
            float lo_t[3], hi_t[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                lo_t[a] = (b[3 * sgn[a] + a] - org[a]) * inv[a];
                hi_t[a] = (b[3 * (1 - sgn[a]) + a] - org[a]) * inv[a];
            }
            bool valid = (lo_t[0] <= hi_t[1]) && (lo_t[1] <= hi_t[0]);
            tmin = lo_t[1] > lo_t[0] ? lo_t[1] : lo_t[0];
            tmax = hi_t[1] < hi_t[0] ? hi_t[1] : hi_t[0];
            valid = valid && (tmin <= hi_t[2]) && (lo_t[2] <= tmax);
            tmin = lo_t[2] > tmin ? lo_t[2] : tmin;
            tmax = hi_t[2] < tmax ? hi_t[2] : tmax;
            return valid && (tmin >= near_) && (tmax <= far_);
        };
        for (int n = lane; n < nvox; n += 64) {
            float tmin, tmax;
            hit[n] = slab(n, tmin, tmax) ? 1 : 0;
            key[n] = tmin;
            perm1[n] = (unsigned short)n;
        }
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) { Introsort<CmpAsc> s1{key, perm1, CmpAsc()}; s1.sort(nvox); }      // tree.py:300
        __builtin_amdgcn_wave_barrier();
        // crossed boxes in sorted order (tree.py:303-309: values placed through the boolean mask, i.e. in order)
        int K = 0;
        for (int base = 0; base < nvox; base += 64) {
            const int p = base + lane;
            bool v = false;
            int n = 0;
            if (p < nvox) { n = perm1[p]; v = hit[n] != 0; }
            const unsigned long long bal = __ballot(v);
            if (v) {
                float tmin, tmax;
                slab(n, tmin, tmax);
                const int pos = K + __popcll(bal & ((1ull << lane) - 1ull));
                r_tmin[pos] = tmin;
                r_cum[pos] = tmax - tmin;
            }
            K += __popcll(bal);
            if (p < nvox) { key[p] = v ? 1.0f : 0.0f; perm2[p] = (unsigned short)p; }
        }
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) { Introsort<CmpDesc> s2{key, perm2, CmpDesc()}; s2.sort(nvox); }    // tree.py:306
        __builtin_amdgcn_wave_barrier();
        double carry = 0.0;
        for (int base = 0; base < K; base += 64) {
            const int i = base + lane;
            double incl = i < K ? (double)r_cum[i] : 0.0;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const double pv = shfl_up_d(incl, off);
                if (lane >= off) incl += pv;
            }
            incl += carry;
            if (i < K) r_cum[i] = (float)incl;
            const int lo = __double2loint(incl), hi = __double2hiint(incl);
            carry = __hiloint2double(__shfl(hi, 63), __shfl(lo, 63));
        }
        __builtin_amdgcn_wave_barrier();
        const float total = K > 0 ? r_cum[K - 1] : 0.0f;
        for (int j = lane; j < samples; j += 64) {
            const float s = u[j] * total;
            int lo = 0, hi = K > 0 ? K - 1 : 0;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (r_cum[mid] < s) lo = mid + 1; else hi = mid;
            }
            p_s[j] = s;
            p_bucket[j] = lo;
        }
        __builtin_amdgcn_wave_barrier();
        for (int j = lane; j < samples; j += 64) {
            const int bkt = p_bucket[j];
            int lo = 0, hi = j;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (p_bucket[mid] < bkt) lo = mid + 1; else hi = mid;
            }
            const float offset = p_s[j] - p_s[lo];
            p_z[j] = (K > 0 ? r_tmin[bkt] : 0.0f) + offset;
            p_vid[j] = perm1[perm2[bkt]];          // tree.py:331-332: crosses_sorted.indices[crosses_start.indices[bucket]]
            p_ix[j] = (unsigned short)j;
        }
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) { Introsort<CmpAsc> s3{p_z, p_ix, CmpAsc()}; s3.sort(samples); }    // tree.py:335
        __builtin_amdgcn_wave_barrier();
        for (int j = lane; j < samples; j += 64) {
            z_out[ray * samples + j] = p_z[j];
            idx_out[ray * samples + j] = p_vid[p_ix[j]];
        }
        if (lane == 0) mask_out[ray] = K > 0 ? 1 : 0;
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- training-time weight integration (tree.py:177-206) ------------------------------------------------------
// acc[v] = sum of the sample weights that fell into voxel v over the whole ray batch, freq[v] = how many of them
// were still visible (mask_weights); the reference materialises two dense (R, N) scatter targets per step, here
// every sample adds into two length-N fp64 accumulators (native fp64 atomics; the fp64 sum rounded to fp32 is
// order-independent in practice), then memm[v] += (acc / freq - memm[v]) / counter where freq > 0.
__global__ void tree_scatter_kernel(const int64_t* __restrict__ idx, const float* __restrict__ w,
                                    const float* __restrict__ mw, int64_t n, int nvox, double* __restrict__ acc,
                                    double* __restrict__ freq) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t v = idx[i];
    if (v < 0 || v >= nvox) return;
    const float wi = w[i], mi = mw[i];
    if (wi != 0.0f) atomicAdd(&acc[v], (double)wi);
    if (mi != 0.0f) atomicAdd(&freq[v], (double)mi);
}

__global__ void tree_update_kernel(const double* __restrict__ acc, const double* __restrict__ freq,
                                   float* __restrict__ memm, int nvox, float counter) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nvox) return;
    const float f = (float)freq[v];
    if (f > 0.0f) {
        const float mean = (float)acc[v] / f;
        const float m = memm[v];
        memm[v] = m + (mean - m) / counter;
    }
}

}  // namespace nm

using namespace nm;

extern "C" int64_t nm_tree_workspace_bytes(int32_t nvox) { return nvox > 0 ? (int64_t)nvox * 16 : 0; }

extern "C" int nm_tree_integrate(const int64_t* d_idx, const float* d_weights, const float* d_mask_weights, int64_t count,
                                 int32_t nvox, int32_t counter, float* d_memm, void* d_workspace, void* stream_) {
    NM_REQUIRE(d_idx && d_weights && d_mask_weights && d_memm && d_workspace, "bad argument");
    NM_REQUIRE(nvox > 0 && counter >= 1 && count >= 0, "tree_integrate: bad sizes");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    double* acc = static_cast<double*>(d_workspace);
    double* freq = acc + nvox;
    NM_HIP_CHECK(hipMemsetAsync(d_workspace, 0, (size_t)nvox * 16, stream));
    if (count > 0)
        hipLaunchKernelGGL(tree_scatter_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, stream, d_idx,
                           d_weights, d_mask_weights, count, nvox, acc, freq);
    hipLaunchKernelGGL(tree_update_kernel, dim3((unsigned)((nvox + 255) / 256)), dim3(256), 0, stream, acc, freq, d_memm,
                       nvox, (float)counter);
    NM_HIP_CHECK(hipGetLastError());
    return 0;
}

// one overflow flag per device (allocated on the device the call runs on)
static int* overflow_flag() {
    static int* flags[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    if (!flags[dev]) {
        if (hipMalloc(&flags[dev], sizeof(int)) != hipSuccess) return nullptr;
        if (hipMemset(flags[dev], 0, sizeof(int)) != hipSuccess) return nullptr;
    }
    return flags[dev];
}

extern "C" int nm_buff_intersect_ex(const float* d_voxels, int32_t nvox, const float* d_origins, int origins_per_ray,
                                    const float* d_dirs, float near_, float far_, const float* d_u, int64_t rays,
                                    int32_t samples, int32_t tie_order, float* d_z, int64_t* d_idx, uint8_t* d_mask,
                                    void* stream_) {
    NM_REQUIRE(d_voxels && d_origins && d_dirs && d_u && d_z && d_idx && d_mask, "bad argument");
    NM_REQUIRE(nvox > 0 && samples > 0 && samples <= BUFF_MAX_SAMPLES, "buff_intersect: samples must be in [1, 512]");
    NM_REQUIRE(tie_order == NM_TIES_STABLE || tie_order == NM_TIES_REFERENCE, "buff_intersect: unknown tie order");
    if (rays <= 0) return 0;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (tie_order == NM_TIES_REFERENCE) {
        NM_REQUIRE(nvox <= 8192, "buff_intersect(reference tie order): at most 8192 voxels");
        const int npad = (nvox + 63) & ~63;
        const size_t lds = (size_t)npad * (3 * 4 + 2 * 2 + 1) + BUFF_MAX_SAMPLES * (3 * 4 + 2 * 2);
        static size_t attr = 0;
        if (attr < lds) {
            NM_HIP_CHECK(hipFuncSetAttribute((const void*)buff_reference_ids_kernel,
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr = lds;
        }
        hipLaunchKernelGGL(buff_reference_ids_kernel, dim3((unsigned)(rays < 8192 ? rays : 8192)), dim3(64), lds, stream,
                           d_voxels, nvox, npad, d_origins, origins_per_ray, d_dirs, near_, far_, d_u, rays, samples, d_z,
                           d_idx, d_mask);
        NM_HIP_CHECK(hipGetLastError());
        return 0;
    }
    int* d_overflow = overflow_flag();
    NM_REQUIRE(d_overflow != nullptr, "buff_intersect: cannot allocate the overflow flag");
    const int64_t blocks = (rays + 3) / 4;
    hipLaunchKernelGGL(buff_intersect_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, stream,
                       d_voxels, nvox, d_origins, origins_per_ray, d_dirs, near_, far_, d_u, rays, samples, d_z, d_idx,
                       d_mask, d_overflow);
    NM_HIP_CHECK(hipGetLastError());
    int h = 0;
    NM_HIP_CHECK(hipMemcpyAsync(&h, d_overflow, sizeof(int), hipMemcpyDeviceToHost, stream));
    NM_HIP_CHECK(hipStreamSynchronize(stream));
    if (h) {
        NM_HIP_CHECK(hipMemset(d_overflow, 0, sizeof(int)));
        set_error("buff_intersect: a ray crosses more than 512 voxels (BUFF_MAX_HITS)");
        return 4;
    }
    return 0;
}

extern "C" int nm_buff_intersect(const float* d_voxels, int32_t nvox, const float* d_origins, int origins_per_ray,
                                 const float* d_dirs, float near_, float far_, const float* d_u, int64_t rays,
                                 int32_t samples, float* d_z, int64_t* d_idx, uint8_t* d_mask, void* stream_) {
    return nm_buff_intersect_ex(d_voxels, nvox, d_origins, origins_per_ray, d_dirs, near_, far_, d_u, rays, samples,
                                NM_TIES_STABLE, d_z, d_idx, d_mask, stream_);
}

extern "C" int nm_buff_intersect_random(const float* d_voxels, int32_t nvox, const float* d_origins, int origins_per_ray,
                                        const float* d_dirs, float near_, float far_, const double* d_u_pick,
                                        const float* d_u_pos, int64_t rays, int32_t samples, float* d_z, int64_t* d_idx,
                                        uint8_t* d_mask, void* stream_) {
    NM_REQUIRE(d_voxels && d_origins && d_dirs && d_u_pick && d_u_pos && d_z && d_idx && d_mask, "bad argument");
    NM_REQUIRE(nvox > 0 && samples > 0 && samples <= BUFF_MAX_SAMPLES, "buff_intersect_random: samples must be in [1, 512]");
    if (rays <= 0) return 0;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    int* d_overflow = overflow_flag();
    NM_REQUIRE(d_overflow != nullptr, "buff_intersect_random: cannot allocate the overflow flag");
    const int64_t blocks = (rays + 3) / 4;
    hipLaunchKernelGGL(buff_random_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, stream, d_voxels,
                       nvox, d_origins, origins_per_ray, d_dirs, near_, far_, d_u_pick, d_u_pos, rays, samples, d_z, d_idx,
                       d_mask, d_overflow);
    NM_HIP_CHECK(hipGetLastError());
    int h = 0;
    NM_HIP_CHECK(hipMemcpyAsync(&h, d_overflow, sizeof(int), hipMemcpyDeviceToHost, stream));
    NM_HIP_CHECK(hipStreamSynchronize(stream));
    if (h) {
        NM_HIP_CHECK(hipMemset(d_overflow, 0, sizeof(int)));
        set_error("buff_intersect_random: a ray crosses more than 512 voxels (BUFF_MAX_HITS)");
        return 4;
    }
    return 0;
}
