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

#include <map>
#include <mutex>
#include <vector>

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
// (c) the order of samples with equal depth.  buff_reference_ids_kernel replays exactly that, one wavefront per ray: the
// ids are the reference's, bit for bit, also on rays where they are not the voxels the samples lie in.  Below first
// the algorithm as libstdc++ writes it (sequential; used for the heap-sort fallback and on the host), then its
// wave-parallel evaluation.
struct CmpAsc {   // KeyValueCompAsc<float>: NaNs last
    __host__ __device__ __forceinline__ bool operator()(float a, float b) const { return (!(a != a) && (b != b)) || (a < b); }
};
struct CmpDesc {  // KeyValueCompDesc<float>: NaNs first
    __host__ __device__ __forceinline__ bool operator()(float a, float b) const { return ((a != a) && !(b != b)) || (a > b); }
};

template <typename Cmp>
struct Introsort {   // libstdc++ bits/stl_algo.h: __sort = __introsort_loop + __final_insertion_sort, threshold 16
    float* k;
    unsigned short* ix;
    Cmp cmp;
    __host__ __device__ __forceinline__ void swap(int i, int j) {
        const float tk = k[i]; k[i] = k[j]; k[j] = tk;
        const unsigned short ti = ix[i]; ix[i] = ix[j]; ix[j] = ti;
    }
    __host__ __device__ void move_median_to_first(int result, int a, int b, int c) {
        if (cmp(k[a], k[b])) {
            if (cmp(k[b], k[c])) swap(result, b);
            else if (cmp(k[a], k[c])) swap(result, c);
            else swap(result, a);
        } else if (cmp(k[a], k[c])) swap(result, a);
        else if (cmp(k[b], k[c])) swap(result, c);
        else swap(result, b);
    }
    __host__ __device__ int unguarded_partition(int first, int last, int pivot) {
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
    __host__ __device__ void push_heap(int first, int hole, int top, float vk, unsigned short vi) {
        int parent = (hole - 1) / 2;
        while (hole > top && cmp(k[first + parent], vk)) {
            k[first + hole] = k[first + parent]; ix[first + hole] = ix[first + parent];
            hole = parent; parent = (hole - 1) / 2;
        }
        k[first + hole] = vk; ix[first + hole] = vi;
    }
    __host__ __device__ void adjust_heap(int first, int hole, int len, float vk, unsigned short vi) {
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
    __host__ __device__ void heap_sort(int first, int last) {
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
    __host__ __device__ void unguarded_linear_insert(int last) {
        const float vk = k[last]; const unsigned short vi = ix[last];
        int next = last - 1;
        while (cmp(vk, k[next])) { k[last] = k[next]; ix[last] = ix[next]; last = next; --next; }
        k[last] = vk; ix[last] = vi;
    }
    __host__ __device__ void insertion_sort(int first, int last) {
        if (first == last) return;
        for (int i = first + 1; i != last; ++i) {
            if (cmp(k[i], k[first])) {
                const float vk = k[i]; const unsigned short vi = ix[i];
                for (int j = i; j > first; --j) { k[j] = k[j - 1]; ix[j] = ix[j - 1]; }
                k[first] = vk; ix[first] = vi;
            } else unguarded_linear_insert(i);
        }
    }
    __host__ __device__ void sort(int n) {
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


// ---- the same three sorts, wave-parallel and still EXACT -----------------------------------------------------------
// Running libstdc++'s introsort on one lane (above, round 2) costs ~1 ms per ray: every compare is an LDS round trip.
// Its result, however, is a function of data-parallel steps:
//  * __unguarded_partition(first, last, pivot): the left scan stops at every element with !(k < pivot) ("left
//    stoppers"), the right scan at every element with !(pivot < k) ("right stoppers"); the j-th swap exchanges the j-th
//    left stopper counted from the left with the j-th right stopper counted from the right, for as long as the former
//    lies in front of the latter.  Neither scan ever revisits a swapped slot before the scans meet, so the pairing is a
//    function of the ORIGINAL stopper positions: a ballot / popcount prefix gives every stopper its ordinal, two position
//    lists in LDS pair them, all swaps of a partition happen at once.  The returned cut is min(next left stopper,
//    last swapped right position) -- see partition() for the case analysis.
//  * __final_insertion_sort is a STABLE sort of what the partition loop leaves, and that array consists of runs of at
//    most 16 elements (longer only where the heap-sort fallback already sorted them) with every run <= the next one: an
//    element's final slot is its position plus (smaller elements among the 15 behind it) minus (larger elements among
//    the 15 in front of it) -- 30 comparisons, no data movement.
//  * only the slots of the crossed boxes are needed from sorts 1 and 2: a range without any of them is left alone
//    (partitioning permutes within its range only) -- for the 0/1 keys of sort 2 that skips almost everything.
// One wavefront per ray, ~22 KB of LDS (7 rays per CU).  The sequential struct above remains as the heap-sort fallback
// (depth limit 2 lg n: adversarial inputs only) and, on the host, for the one data-independent case -- a ray that
// crosses nothing sorts an all-equal mask, whose first slot depends on n alone.
constexpr int REF_MAX_HITS = 512;   // boxes one ray may cross in this mode; rays are first tried with room for REF_FAST_HITS
constexpr int REF_FAST_HITS = 128;
constexpr int EQ_MIN = 17, EQ_MAX = 128;   // all-equal ranges of these lengths are permuted from a table (see WaveSort::loop)

template <typename Cmp>
struct WaveSort {
    float* k;
    unsigned short* ix;
    unsigned short* la;        // left stoppers, ascending, first len/2 + 2 of them
    unsigned short* lb;        // right stoppers, ascending (the j-th from the right is lb[nr - 1 - j])
    unsigned short* stack;     // pending ranges: (first, last, depth) triples
    int lane;
    Cmp cmp;
    const unsigned char* eq_table;   // eq_table[eq_offset(len) + i]: source index of slot i after sorting len equal keys

    __device__ __forceinline__ int uni(int v) const { return __builtin_amdgcn_readfirstlane(v); }

    // __unguarded_partition over [first, last) against k[pivot]; every lane takes part
    __device__ int partition(int first, int last, int pivot) {
        const float pk = k[pivot];
        const int cap = (last - first) / 2 + 2;
        int nl = 0, nr = 0;
        const unsigned long long below = (1ull << lane) - 1ull;
        for (int base = first; base < last; base += 64) {
            const int p = base + lane;
            bool L = false, R = false;
            if (p < last) { const float v = k[p]; L = !cmp(v, pk); R = !cmp(pk, v); }
            const unsigned long long bl = __ballot(L), br = __ballot(R);
            if (L) { const int j = nl + __popcll(bl & below); if (j < cap) la[j] = (unsigned short)p; }
            if (R) lb[nr + __popcll(br & below)] = (unsigned short)p;
            nl += __popcll(bl); nr += __popcll(br);
        }
        __builtin_amdgcn_wave_barrier();
        const int lim = min(min(nl, cap), nr);
        int J = 0;
        for (int base = 0; base < lim; base += 64) {
            const int j = base + lane;
            int a = 0, b = 0;
            bool ok = false;
            if (j < lim) { a = la[j]; b = lb[nr - 1 - j]; ok = a < b; }      // true for a prefix of j: a ascends, b descends
            const int c = __popcll(__ballot(ok));
            if (ok) {
                const float ka = k[a], kb = k[b]; k[a] = kb; k[b] = ka;
                const unsigned short ia = ix[a], ib = ix[b]; ix[a] = ib; ix[b] = ia;
            }
            J += c;
            if (c < 64) break;
        }
        __builtin_amdgcn_wave_barrier();
        // The scans meet after J swaps.  The left scan resumes behind its last swap and stops at the next left stopper
        // of the CURRENT array: the next original one if it lies in front of the last swapped right position b_{J-1}
        // (everything between the last swapped pair is untouched), else b_{J-1} itself, which now holds a left stopper.
        // The right scan then stops at or in front of it (an original right stopper in the untouched middle, the last
        // swapped left position, or the pivot), so the loop exits and returns that left position.
        const int a_next = (J < nl) ? (int)la[J] : 0x7fffffff;              // J <= (len - 1) / 2 < cap
        const int b_last = (J > 0) ? (int)lb[nr - J] : 0x7fffffff;
        return uni(min(a_next, b_last));
    }

    template <typename Keep>   // keep(first, last): does the range hold an element whose final slot is needed?
    __device__ void loop(int n, Keep keep) {
        if (n <= 16) return;
        int lg = 0;
        while ((2 << lg) <= n) ++lg;
        int sp = 0, first = 0, last = n, depth = 2 * lg;
        for (;;) {
            while (last - first > 16) {
                if (!keep(first, last)) break;
                // A range of equivalent keys sorts data-independently (every comparison is false): the partitions reverse
                // and halve it the same way whatever it holds, so the whole sub-sort is one gather through a table built
                // by the sequential algorithm on the host.  Ties are the rule here: every box entered through the same
                // plane shares its entry depth.
                const int len = last - first;
                if (eq_table && len <= EQ_MAX && depth >= 4) {
                    const float v0 = k[first];
                    bool diff = false;
                    float kv[2];
                    unsigned short iv[2];
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int p = first + q * 64 + lane;
                        kv[q] = p < last ? k[p] : v0;
                        diff = diff || cmp(kv[q], v0) || cmp(v0, kv[q]);
                    }
                    if (__ballot(diff) == 0ull) {
                        const unsigned char* tab = eq_table + (len * (len - 1) / 2 - EQ_MIN * (EQ_MIN - 1) / 2);
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const int i = q * 64 + lane;
                            if (i < len) { const int src = first + tab[i]; kv[q] = k[src]; iv[q] = ix[src]; }
                        }
                        __builtin_amdgcn_wave_barrier();
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const int i = q * 64 + lane;
                            if (i < len) { k[first + i] = kv[q]; ix[first + i] = iv[q]; }
                        }
                        __builtin_amdgcn_wave_barrier();
                        break;
                    }
                }
                if (depth == 0) {
                    if (lane == 0) { Introsort<Cmp> seq{k, ix, cmp}; seq.heap_sort(first, last); }
                    __builtin_amdgcn_wave_barrier();
                    break;
                }
                --depth;
                const int mid = first + (last - first) / 2, a = first + 1, c = last - 1;
                const float ka = k[a], kb = k[mid], kc = k[c];
                int m;                                     // __move_median_to_first(first, a, mid, c)
                if (cmp(ka, kb)) m = cmp(kb, kc) ? mid : (cmp(ka, kc) ? c : a);
                else m = cmp(ka, kc) ? a : (cmp(kb, kc) ? c : mid);
                m = uni(m);
                if (lane == 0) {
                    const float t = k[first]; k[first] = k[m]; k[m] = t;
                    const unsigned short ti = ix[first]; ix[first] = ix[m]; ix[m] = ti;
                }
                __builtin_amdgcn_wave_barrier();
                const int cut = partition(first + 1, last, first);
                if (lane == 0) { stack[3 * sp] = (unsigned short)cut; stack[3 * sp + 1] = (unsigned short)last; stack[3 * sp + 2] = (unsigned short)depth; }
                ++sp;
                last = cut;
            }
            if (sp == 0) break;
            --sp;
            __builtin_amdgcn_wave_barrier();
            first = uni(stack[3 * sp]); last = uni(stack[3 * sp + 1]); depth = uni(stack[3 * sp + 2]);
        }
        __builtin_amdgcn_wave_barrier();
    }

    // slot of element i after the final (stable) insertion sort
    __device__ __forceinline__ int final_slot(int i, int n) const {
        const float v = k[i];
        int f = i;
        const int hi = min(n - 1, i + 15), lo = max(0, i - 15);
        for (int j = i + 1; j <= hi; ++j) f += cmp(k[j], v) ? 1 : 0;
        for (int j = lo; j < i; ++j) f -= cmp(v, k[j]) ? 1 : 0;
        return f;
    }
};

template <int CAP>
__global__ __launch_bounds__(64) void buff_reference_ids_kernel(const float* __restrict__ voxels, int nvox, int npad, int spad,
                                                               int key_bytes, int la_cap, int list_bytes, int zero_slot0,
                                                               const unsigned char* __restrict__ eq_table,
                                                               const float* __restrict__ origins, int origins_per_ray,
                                                               const float* __restrict__ dirs, float near_, float far_,
                                                               const float* __restrict__ u, int64_t rays, int samples,
                                                               float* __restrict__ z_out, int64_t* __restrict__ idx_out,
                                                               uint8_t* __restrict__ mask_out, int* __restrict__ overflow) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // region A (key_bytes): sort keys; later the per-sample arrays.  region B (list_bytes): stopper lists / f per crossed box
    float* key = reinterpret_cast<float*>(smem);
    unsigned short* la = reinterpret_cast<unsigned short*>(smem + key_bytes);
    unsigned short* lb = la + la_cap;
    unsigned short* perm = reinterpret_cast<unsigned short*>(smem + key_bytes + list_bytes);    // [npad] index | hit << 15; later markers
    unsigned short* T = perm + npad;                                   // [CAP] box id of the c-th crossed box
    unsigned short* vslot = T + CAP;                          // [CAP] box id attributed to bucket b
    unsigned short* stack = vslot + CAP;                      // [96]
    float* r_tmin = reinterpret_cast<float*>(stack + 96);              // [CAP] crossed boxes in reference order
    float* r_cum = r_tmin + CAP;
    // per-sample arrays (alias region A once the sorts over the boxes are done)
    float* p_s = key;
    float* p_z = p_s + spad;
    unsigned short* p_bucket = reinterpret_cast<unsigned short*>(p_z + spad);
    unsigned short* p_ix = p_bucket + spad;
    unsigned short* p_vid = p_ix + spad;
    unsigned short* fslot = la;                                        // [CAP] final slot (sort 1) of the c-th crossed box
    const int lane = threadIdx.x;
    const unsigned long long below = (1ull << lane) - 1ull;
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
        int K = 0;
        for (int base = 0; base < nvox; base += 64) {
            const int n = base + lane;
            bool h = false;
            if (n < nvox) {
                float tmin, tmax;
                h = slab(n, tmin, tmax);
                key[n] = tmin;
                perm[n] = (unsigned short)(n | (h ? 0x8000 : 0));
            }
            K += __popcll(__ballot(h));
        }
        if (K > CAP) { if (lane == 0) atomicExch(overflow, 1); K = CAP; }
        __builtin_amdgcn_wave_barrier();
        // ---- sort 1 (tree.py:300): entry depths ascending.  Only the crossed boxes' slots reach the output, so ranges of
        // ANY size without a crossed box are left alone (their entry depths lie in a band: after two or three levels most
        // ranges hold none -- checking the large ranges too took the kernel from 4.7 to 4.0 ms).  A ray that crosses nothing
        // reports the box in one particular slot of the sorted order: then only the ranges that can still reach that slot
        // (its +-15 insertion window, and the +-15 around each of those) are sorted -- a selection, not a sort.
        WaveSort<CmpAsc> s1{key, perm, la, lb, stack, lane, CmpAsc(), eq_table};
        const bool none = K == 0;
        s1.loop(nvox, [&](int first, int last) -> bool {
            if (none) return first <= zero_slot0 + 30 && last > zero_slot0 - 30;
            bool any = false;
            for (int base = first; base < last; base += 64) { const int p = base + lane; any = any || (p < last && (perm[p] & 0x8000)); }
            return __ballot(any) != 0ull;
        });
        int lone = 0;                                     // K == 0: the box in slot zero_slot0 of the sorted order
        if (none) {
            const int i = zero_slot0 - 15 + lane;
            const bool mine = lane < 31 && i >= 0 && i < nvox && s1.final_slot(i, nvox) == zero_slot0;
            const unsigned long long bm = __ballot(mine);
            const int src = bm ? __builtin_ctzll(bm) : 0;
            lone = __shfl(i >= 0 && i < nvox ? (int)(perm[i] & 0x7fff) : 0, src);
        }
        // ---- the crossed boxes: id, final slot of sort 1 (= position in the sorted order)
        int seen = 0;
        for (int base = 0; base < nvox && !none; base += 64) {
            const int i = base + lane;
            const bool h = i < nvox && (perm[i] & 0x8000);
            const unsigned long long bal = __ballot(h);
            if (h) {
                const int c = seen + __popcll(bal & below);
                if (c < CAP) { T[c] = perm[i] & 0x7fff; fslot[c] = (unsigned short)s1.final_slot(i, nvox); }
            }
            seen += __popcll(bal);
        }
        __builtin_amdgcn_wave_barrier();
        // crossed boxes in sorted order (tree.py:303-309: values placed through the boolean mask, i.e. in order)
        for (int c = lane; c < K; c += 64) {
            const int f = fslot[c];
            int pos = 0;
            for (int e = 0; e < K; ++e) pos += fslot[e] < f ? 1 : 0;
            float tmin, tmax;
            slab(T[c], tmin, tmax);
            r_tmin[pos] = tmin;
            r_cum[pos] = tmax - tmin;
        }
        // ---- sort 2 (tree.py:306): the 0/1 hit sequence of the sorted order, descending; the element in slot f carries
        // the ordinal of its crossed box
        int myf[CAP / 64];
#pragma unroll
        for (int q = 0; q < CAP / 64; ++q) myf[q] = (q * 64 + lane < K) ? (int)fslot[q * 64 + lane] : -1;
        __builtin_amdgcn_wave_barrier();
        for (int p = lane; p < nvox; p += 64) { key[p] = 0.0f; perm[p] = 0; }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < CAP / 64; ++q)
            if (myf[q] >= 0) { key[myf[q]] = 1.0f; perm[myf[q]] = (unsigned short)(q * 64 + lane); }
        __builtin_amdgcn_wave_barrier();
        WaveSort<CmpDesc> s2{key, perm, la, lb, stack, lane, CmpDesc(), nullptr};
        s2.loop(nvox, [&](int first, int last) -> bool {
            bool any = false;
            for (int base = first; base < last; base += 64) { const int p = base + lane; any = any || (p < last && key[p] != 0.0f); }
            return __ballot(any) != 0ull;
        });
        // the final insertion sort is stable: the ones take the first K slots in the order of their positions
        seen = 0;
        for (int base = 0; base < nvox && !none; base += 64) {
            const int p = base + lane;
            const bool one = p < nvox && key[p] != 0.0f;
            const unsigned long long bal = __ballot(one);
            if (one) vslot[seen + __popcll(bal & below)] = T[perm[p]];
            seen += __popcll(bal);
        }
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
            p_bucket[j] = (unsigned short)lo;
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
            p_vid[j] = (unsigned short)(K > 0 ? (int)vslot[bkt] : lone);   // tree.py:331-332: crosses_sorted.indices[crosses_start.indices[bucket]]
            p_ix[j] = (unsigned short)j;
        }
        __builtin_amdgcn_wave_barrier();
        // ---- sort 3 (tree.py:335): the sample depths ascending, every slot needed
        // The depths are placed bucket by bucket along the ray, so they usually arrive strictly increasing -- and a sequence
        // of DISTINCT keys has one sorted order whatever the algorithm (an unstable sort only decides ties): the sort is
        // then the identity.  Any tie, inversion or NaN takes the real thing.
        bool rising = true;
        for (int j = lane; j + 1 < samples; j += 64) rising = rising && (p_z[j] < p_z[j + 1]);
        if (__ballot(!rising) == 0ull) {
            for (int j = lane; j < samples; j += 64) {
                z_out[ray * samples + j] = p_z[j];
                idx_out[ray * samples + j] = p_vid[j];
            }
        } else {
            WaveSort<CmpAsc> s3{p_z, p_ix, la, lb, stack, lane, CmpAsc(), eq_table};
            s3.loop(samples, [](int, int) -> bool { return true; });
            for (int j = lane; j < samples; j += 64) {
                const int f = s3.final_slot(j, samples);
                z_out[ray * samples + f] = p_z[j];
                idx_out[ray * samples + f] = p_vid[p_ix[j]];
            }
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

// The reference's second sort on a ray that crosses nothing orders an all-equal mask: data-independent, so the index
// that ends up in slot 0 is a function of n alone (the same libstdc++ algorithm, run once per n on the host).
static int all_equal_first_slot(int n) {
    static std::mutex lock;
    static std::map<int, int> cache;
    std::lock_guard<std::mutex> g(lock);
    auto it = cache.find(n);
    if (it != cache.end()) return it->second;
    std::vector<float> k((size_t)n, 0.0f);
    std::vector<unsigned short> ix((size_t)n);
    for (int i = 0; i < n; ++i) ix[(size_t)i] = (unsigned short)i;
    Introsort<CmpDesc> s{k.data(), ix.data(), CmpDesc()};
    s.sort(n);
    return cache[n] = (int)ix[0];
}

// Sorting `len` equivalent keys is data-independent: slot i of the result holds the element that stood at
// eq_table[offset(len) + i].  Built once per device with the sequential algorithm (lengths EQ_MIN..EQ_MAX, 8 KB).
static const unsigned char* equal_keys_table() {
    static std::mutex lock;
    static unsigned char* tables[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> g(lock);
    if (tables[dev]) return tables[dev];
    std::vector<unsigned char> host;
    for (int len = EQ_MIN; len <= EQ_MAX; ++len) {
        std::vector<float> k((size_t)len, 0.0f);
        std::vector<unsigned short> ix((size_t)len);
        for (int i = 0; i < len; ++i) ix[(size_t)i] = (unsigned short)i;
        Introsort<CmpAsc> s{k.data(), ix.data(), CmpAsc()};
        s.sort(len);
        for (int i = 0; i < len; ++i) host.push_back((unsigned char)ix[(size_t)i]);
    }
    unsigned char* d = nullptr;
    if (hipMalloc(&d, host.size()) != hipSuccess) return nullptr;
    if (hipMemcpy(d, host.data(), host.size(), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(d); return nullptr; }
    return tables[dev] = d;
}

// one overflow flag per device (allocated on the device the call runs on)
static int* overflow_flag() {
    static int* flags[64] = {};
    static std::mutex lock;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> g(lock);
    if (!flags[dev]) {
        int* f = nullptr;
        if (hipMalloc(&f, sizeof(int)) != hipSuccess) return nullptr;
        if (hipMemset(f, 0, sizeof(int)) != hipSuccess) { (void)hipFree(f); return nullptr; }
        flags[dev] = f;
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
        const int npad = (nvox + 63) & ~63, spad = (samples + 63) & ~63, lpad = npad > spad ? npad : spad;
        const int key_bytes = ((4 * npad > 14 * spad ? 4 * npad : 14 * spad) + 15) & ~15;
        int* d_over = overflow_flag();
        NM_REQUIRE(d_over != nullptr, "buff_intersect: cannot allocate the overflow flag");
        const unsigned char* d_eq = equal_keys_table();
        NM_REQUIRE(d_eq != nullptr, "buff_intersect: cannot allocate the equal-keys table");
        const int zero_slot0 = all_equal_first_slot(nvox);
        int dev = 0;
        NM_HIP_CHECK(hipGetDevice(&dev));
        // rays are first run with room for REF_FAST_HITS crossed boxes each (less LDS: more rays in flight); if one of them
        // crosses more, the whole call is repeated with REF_MAX_HITS
        for (int pass = 0; pass < 2; ++pass) {
            const int cap = pass == 0 ? REF_FAST_HITS : REF_MAX_HITS;
            const int la_cap = lpad / 2 + 64 > cap ? lpad / 2 + 64 : cap;
            const int list_bytes = (2 * (la_cap + lpad) + 15) & ~15;
            const size_t lds = (size_t)key_bytes + list_bytes + 2 * npad + 2 * 2 * cap + 2 * 96 + 2 * 4 * cap;
            const void* fn = pass == 0 ? (const void*)buff_reference_ids_kernel<REF_FAST_HITS> : (const void*)buff_reference_ids_kernel<REF_MAX_HITS>;
            static size_t attr[2][64] = {};               // hipFuncAttributeMaxDynamicSharedMemorySize is per device
            static std::mutex attr_lock;
            {
                std::lock_guard<std::mutex> g(attr_lock);
                if (dev < 0 || dev >= 64 || attr[pass][dev] < lds) {
                    NM_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                    if (dev >= 0 && dev < 64) attr[pass][dev] = lds;
                }
            }
            // persistent, one wavefront per ray in flight, EXACTLY as many as are resident: the kernel is latency-bound (its
            // time is inversely proportional to the rays in flight: 2 / 4 / 6 / 9 per CU -> 20.9 / 11.6 / 7.9 / 5.7 ms), and a
            // grid a third larger than the residency (round 3's first version) ran 7.6 ms -- a second, mostly empty round.
            // (More rays in flight would help further -- a 512-voxel tree, 6 KB per ray: 9 / 12 / 16 / 24 per CU -> 2.76 / 2.35 /
            // 2.03 / 1.94 ms -- but keeping only the rightmost len / 2 + 2 right stoppers in a ring, 17.0 -> 15.2 KB per ray,
            // did not get a tenth ray onto a CU and cost 5 % in the partition loop: not kept.  Nor was a register-resident
            // version of the introsort loop for ranges of <= 64 elements (stopper lists as ballot masks, j-th stopper by bit
            // select, swaps as cross-lane reads: bit-exact, 5.98 ms against 5.72): the per-call cost is the length of the
            // dependent instruction chain, not the LDS round trips.  Phase shares (wall-clock counters): box test 6 %,
            // sort 1 48 %, ranking 6 %, sort 2 18 %, sample placement 4 %, sort 3 18 %.)
            int cus = 0, lds_cu = 0;
            if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
            if (hipDeviceGetAttribute(&lds_cu, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, dev) != hipSuccess || lds_cu < 65536) lds_cu = 65536;
            const size_t granule = (lds + 1279) / 1280 * 1280;    // LDS is handed out in 1280-byte granules on gfx950 (160 KB / 128)
            int per_cu = (int)((size_t)lds_cu / granule);
            per_cu = per_cu < 1 ? 1 : (per_cu > 32 ? 32 : per_cu);
            const int64_t want = (int64_t)cus * per_cu;
            const dim3 grid((unsigned)(rays < want ? rays : want));
            if (pass == 0)
                hipLaunchKernelGGL(buff_reference_ids_kernel<REF_FAST_HITS>, grid, dim3(64), lds, stream, d_voxels, nvox, npad, spad,
                                   key_bytes, la_cap, list_bytes, zero_slot0, d_eq, d_origins, origins_per_ray, d_dirs, near_, far_,
                                   d_u, rays, samples, d_z, d_idx, d_mask, d_over);
            else
                hipLaunchKernelGGL(buff_reference_ids_kernel<REF_MAX_HITS>, grid, dim3(64), lds, stream, d_voxels, nvox, npad, spad,
                                   key_bytes, la_cap, list_bytes, zero_slot0, d_eq, d_origins, origins_per_ray, d_dirs, near_, far_,
                                   d_u, rays, samples, d_z, d_idx, d_mask, d_over);
            NM_HIP_CHECK(hipGetLastError());
            int h = 0;
            NM_HIP_CHECK(hipMemcpyAsync(&h, d_over, sizeof(int), hipMemcpyDeviceToHost, stream));
            NM_HIP_CHECK(hipStreamSynchronize(stream));
            if (!h) return 0;
            NM_HIP_CHECK(hipMemset(d_over, 0, sizeof(int)));
        }
        set_error("buff_intersect(reference tie order): a ray crosses more than 512 voxels");
        return 4;
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
