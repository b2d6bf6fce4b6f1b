// Lewiner MC33 marching cubes for gfx950, output-identical to skimage.measure.marching_cubes(volume, level)
// (the call at /root/reference/src/mesh_nerf.py:79: method='lewiner', step_size=1, 'descent', no mask):
// same vertices (bit for bit), same vertex NUMBERING, same triangle order, same normals and values.
//
// The CPU algorithm is a sequential scan (axis0 outermost, axis2 innermost) with a per-layer vertex
// cache; vertex ids are "order of first creation", triangle order is "cube scan order, tiling order".
// Parallel restatement used here:
//   1. classify   one thread per cube: corner signs -> case/config -> (face / interior tests) -> tiling row
//                 (offset into the LUT, triangle count) and the number of vertices this cube CREATES.
//                 An edge vertex is created by the first cube in scan order that contains the edge; every
//                 tiling of a cube uses exactly its sign-changing edges, so ownership is a pure function of
//                 the edge position ("no earlier cube contains it").
//                 Sign patterns are computed by every lane (phase A, brick-shaped workgroups marching in z); the ~1 %
//                 of cubes the surface cuts go through an LDS queue so the MC33 tests run densely (phase B).
//                 Per-cube state: ONE byte.
//   2. scan       exclusive prefix sums of (created vertices, triangles, active cubes) over 1024-cube tiles in
//                 scan order (1024-tile groups scanned in parallel, then the group totals).
//   3. compact    active cubes -> list (cube id, vertex base, triangle base), still in scan order.
//   4. vertices   one thread per ACTIVE cube walks its tiling; every first use of an owned edge (or of the
//                 centre vertex, edge id 12) creates vertex base+k: position (fp64 inverse-|v| interpolation
//                 rounded to fp32, as skimage), index stored in an edge->vertex table (int32 volume per axis).
//      faces      one thread per active cube walks its tiling again and looks the indices up; triples are
//                 reversed ('descent').
//   5. normals    one thread per vertex: re-plays, in scan order, the <= 4 cubes sharing its edge and
//                 accumulates their gradient contributions in fp32 in exactly skimage's order, takes the
//                 max of the cubes' value ranges, normalises in fp64.
// HBM-bound integer/byte work: 4 B/voxel streamed once for classification is the algorithmic traffic
// (442 MB at 480^3); the other passes touch only the ~1 % of cubes that are active.
#include <math.h>

#include "nm_internal.h"
#include "mc_luts.h"

namespace nm {

constexpr double SK_EPS = 2.220446049250313e-16;   // skimage's "FLT_EPSILON" is np.spacing(1.0)
constexpr int MC_BLOCK = 256;
constexpr int MC_ITEMS = 4;                          // CONSECUTIVE cubes per thread (their 4 code bytes = one dword)
constexpr int MC_TILE = MC_BLOCK * MC_ITEMS;

struct McDims {
    int n0, n1, n2;          // volume extents (axis0 = skimage z, axis2 = skimage x)
    int c0, c1, c2;          // cube extents (n-1)
    int64_t cubes;
};

__device__ __forceinline__ int lut(int i) { return MC_LUT[i]; }

struct Cube {
    double v[8];             // corner value - iso, Lewiner corner numbering
};

__device__ __forceinline__ void load_cube(const float* __restrict__ vol, const McDims& d, int z, int y, int x,
                                          double iso, Cube& c) {
    const int64_t s1 = d.n2, s0 = (int64_t)d.n1 * d.n2;
    const float* p = vol + (int64_t)z * s0 + (int64_t)y * s1 + x;
    c.v[0] = (double)p[0] - iso;           c.v[1] = (double)p[1] - iso;
    c.v[2] = (double)p[s1 + 1] - iso;      c.v[3] = (double)p[s1] - iso;
    c.v[4] = (double)p[s0] - iso;          c.v[5] = (double)p[s0 + 1] - iso;
    c.v[6] = (double)p[s0 + s1 + 1] - iso; c.v[7] = (double)p[s0 + s1] - iso;
}

// c.v[i] for a run-time i as a select chain over REGISTER copies.  The empty asm makes each copy an opaque value:
// without it the compiler rewrites select(load, load) as load(select(address, address)), keeps the cube in scratch
// memory to index it, and every kernel that runs the MC33 tests pays for scratch (mc_vertex_attributes: 305 -> 84 us).
__device__ __forceinline__ double pick(const Cube& c, int i) {
    double a[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        a[k] = c.v[k];
        asm("" : "+v"(a[k]));
    }
    double r = a[0];
#pragma unroll
    for (int k = 1; k < 8; ++k) r = i == k ? a[k] : r;
    return r;
}

// Lewiner's test_face.  All six faces' (A, A*C - B*D) are formed from statically indexed corners and the requested one
// is selected afterwards: a switch over the corners (or a select chain over c.v[]) is turned into an indexed load by
// the compiler, which moves the cube -- and every kernel that runs the MC33 tests -- to scratch memory.
__device__ __forceinline__ bool test_face(const Cube& c, int face) {
    const double* v = c.v;
    const int f = face < 0 ? -face : face;
    double A = v[4], acbd = v[4] * v[6] - v[7] * v[5];                       // face 6 (and anything else)
    if (f == 1) { A = v[0]; acbd = v[0] * v[5] - v[4] * v[1]; }
    if (f == 2) { A = v[1]; acbd = v[1] * v[6] - v[5] * v[2]; }
    if (f == 3) { A = v[2]; acbd = v[2] * v[7] - v[6] * v[3]; }
    if (f == 4) { A = v[3]; acbd = v[3] * v[4] - v[7] * v[0]; }
    if (f == 5) { A = v[0]; acbd = v[0] * v[2] - v[3] * v[1]; }
    if (acbd > -SK_EPS && acbd < SK_EPS) return face >= 0;
    return face * A * acbd >= 0;
}

// reference edge (p,q) and the three parallel edges (B, C, D) used by the interior test
__device__ __constant__ signed char MC_INTERIOR_EDGES[12][8] = {
    {0, 1, 3, 2, 7, 6, 4, 5}, {1, 2, 0, 3, 4, 7, 5, 6}, {2, 3, 1, 0, 5, 4, 6, 7}, {3, 0, 2, 1, 6, 5, 7, 4},
    {4, 5, 7, 6, 3, 2, 0, 1}, {5, 6, 4, 7, 0, 3, 1, 2}, {6, 7, 5, 4, 1, 0, 2, 3}, {7, 4, 6, 5, 2, 1, 3, 0},
    {0, 4, 3, 7, 2, 6, 1, 5}, {1, 5, 0, 4, 3, 7, 2, 6}, {2, 6, 1, 5, 0, 4, 3, 7}, {3, 7, 2, 6, 1, 5, 0, 4}};

__device__ __forceinline__ bool test_interior(const Cube& c, int kase, int cfg, int subcfg, int s) {   // inlined: a call would pin the cube in scratch
    const double* v = c.v;
    double t, At = 0, Bt, Ct, Dt;
    if (kase == 4 || kase == 10) {
        const double a = (v[4] - v[0]) * (v[6] - v[2]) - (v[7] - v[3]) * (v[5] - v[1]);
        const double b = v[2] * (v[4] - v[0]) + v[0] * (v[6] - v[2]) - v[1] * (v[7] - v[3]) - v[3] * (v[5] - v[1]);
        t = -b / (2 * a + SK_EPS);
        if (t < 0 || t > 1) return s > 0;
        At = v[0] + (v[4] - v[0]) * t;
        Bt = v[3] + (v[7] - v[3]) * t;
        Ct = v[2] + (v[6] - v[2]) * t;
        Dt = v[1] + (v[5] - v[1]) * t;
    } else {
        int edge;
        if (kase == 6) edge = lut(MC_TEST6_OFF + cfg * MC_TEST6_ROW + 2);
        else if (kase == 7) edge = lut(MC_TEST7_OFF + cfg * MC_TEST7_ROW + 4);
        else if (kase == 12) edge = lut(MC_TEST12_OFF + cfg * MC_TEST12_ROW + 3);
        else edge = lut(MC_TILING13_5_1_OFF + (cfg * MC_TILING13_5_1_SUB + subcfg) * MC_TILING13_5_1_ROW);
        const signed char* e = MC_INTERIOR_EDGES[edge];
        const double p0 = pick(c, e[0]), p1 = pick(c, e[1]), p2 = pick(c, e[2]), p3 = pick(c, e[3]);
        const double p4 = pick(c, e[4]), p5 = pick(c, e[5]), p6 = pick(c, e[6]), p7 = pick(c, e[7]);
        t = p0 / (p0 - p1 + SK_EPS);
        Bt = p2 + (p3 - p2) * t;
        Ct = p4 + (p5 - p4) * t;
        Dt = p6 + (p7 - p6) * t;
    }
    const int test = (At >= 0 ? 1 : 0) + (Bt >= 0 ? 2 : 0) + (Ct >= 0 ? 4 : 0) + (Dt >= 0 ? 8 : 0);
    switch (test) {
        case 0: case 1: case 2: case 3: case 4: case 6: case 8: case 9: case 12: return s > 0;
        case 5: return (At * Ct - Bt * Dt < SK_EPS) ? (s > 0) : false;     // skimage returns 0 when the
        case 10: return (At * Ct - Bt * Dt >= SK_EPS) ? (s > 0) : false;   // saddle test fails (not s < 0)
        default: return s < 0;
    }
}

#define MC_ROW(NAME, cfg) (MC_##NAME##_OFF + (cfg) * MC_##NAME##_ROW)
#define MC_ROW3(NAME, cfg, sub) (MC_##NAME##_OFF + ((cfg) * MC_##NAME##_SUB + (sub)) * MC_##NAME##_ROW)
#define MC_T1(NAME, cfg) lut(MC_##NAME##_OFF + (cfg))
#define MC_T2(NAME, cfg, k) lut(MC_##NAME##_OFF + (cfg) * MC_##NAME##_ROW + (k))

// Lewiner's process_cube: -> tiling row offset in MC_LUT and triangle count (0 = empty cube)
__device__ __forceinline__ void select_tiling(const Cube& c, int index, int& offset, int& nt) {
    const int kase = lut(MC_CASES_OFF + 2 * index), cfg = lut(MC_CASES_OFF + 2 * index + 1);
    offset = 0; nt = 0;
    int sub = 0;
#define MC_PICK(off, n) do { offset = (off); nt = (n); } while (0)
    switch (kase) {
        case 1: MC_PICK(MC_ROW(TILING1, cfg), 1); break;
        case 2: MC_PICK(MC_ROW(TILING2, cfg), 2); break;
        case 3:
            if (test_face(c, MC_T1(TEST3, cfg))) MC_PICK(MC_ROW(TILING3_2, cfg), 4);
            else MC_PICK(MC_ROW(TILING3_1, cfg), 2);
            break;
        case 4:
            if (test_interior(c, kase, cfg, 0, MC_T1(TEST4, cfg))) MC_PICK(MC_ROW(TILING4_1, cfg), 2);
            else MC_PICK(MC_ROW(TILING4_2, cfg), 6);
            break;
        case 5: MC_PICK(MC_ROW(TILING5, cfg), 3); break;
        case 6:
            if (test_face(c, MC_T2(TEST6, cfg, 0))) MC_PICK(MC_ROW(TILING6_2, cfg), 5);
            else if (test_interior(c, kase, cfg, 0, MC_T2(TEST6, cfg, 1))) MC_PICK(MC_ROW(TILING6_1_1, cfg), 3);
            else MC_PICK(MC_ROW(TILING6_1_2, cfg), 9);
            break;
        case 7:
            if (test_face(c, MC_T2(TEST7, cfg, 0))) sub += 1;
            if (test_face(c, MC_T2(TEST7, cfg, 1))) sub += 2;
            if (test_face(c, MC_T2(TEST7, cfg, 2))) sub += 4;
            switch (sub) {
                case 0: MC_PICK(MC_ROW(TILING7_1, cfg), 3); break;
                case 1: MC_PICK(MC_ROW3(TILING7_2, cfg, 0), 5); break;
                case 2: MC_PICK(MC_ROW3(TILING7_2, cfg, 1), 5); break;
                case 3: MC_PICK(MC_ROW3(TILING7_3, cfg, 0), 9); break;
                case 4: MC_PICK(MC_ROW3(TILING7_2, cfg, 2), 5); break;
                case 5: MC_PICK(MC_ROW3(TILING7_3, cfg, 1), 9); break;
                case 6: MC_PICK(MC_ROW3(TILING7_3, cfg, 2), 9); break;
                default:
                    if (test_interior(c, kase, cfg, 0, MC_T2(TEST7, cfg, 3))) MC_PICK(MC_ROW(TILING7_4_2, cfg), 9);
                    else MC_PICK(MC_ROW(TILING7_4_1, cfg), 5);
            }
            break;
        case 8: MC_PICK(MC_ROW(TILING8, cfg), 2); break;
        case 9: MC_PICK(MC_ROW(TILING9, cfg), 4); break;
        case 10:
            if (test_face(c, MC_T2(TEST10, cfg, 0))) {
                if (test_face(c, MC_T2(TEST10, cfg, 1))) MC_PICK(MC_ROW(TILING10_1_1_, cfg), 4);
                else MC_PICK(MC_ROW(TILING10_2, cfg), 8);
            } else {
                if (test_face(c, MC_T2(TEST10, cfg, 1))) MC_PICK(MC_ROW(TILING10_2_, cfg), 8);
                else if (test_interior(c, kase, cfg, 0, MC_T2(TEST10, cfg, 2))) MC_PICK(MC_ROW(TILING10_1_1, cfg), 4);
                else MC_PICK(MC_ROW(TILING10_1_2, cfg), 8);
            }
            break;
        case 11: MC_PICK(MC_ROW(TILING11, cfg), 4); break;
        case 12:
            if (test_face(c, MC_T2(TEST12, cfg, 0))) {
                if (test_face(c, MC_T2(TEST12, cfg, 1))) MC_PICK(MC_ROW(TILING12_1_1_, cfg), 4);
                else MC_PICK(MC_ROW(TILING12_2, cfg), 8);
            } else {
                if (test_face(c, MC_T2(TEST12, cfg, 1))) MC_PICK(MC_ROW(TILING12_2_, cfg), 8);
                else if (test_interior(c, kase, cfg, 0, MC_T2(TEST12, cfg, 2))) MC_PICK(MC_ROW(TILING12_1_1, cfg), 4);
                else MC_PICK(MC_ROW(TILING12_1_2, cfg), 8);
            }
            break;
        case 13: {
            for (int k = 0; k < 6; ++k)
                if (test_face(c, MC_T2(TEST13, cfg, k))) sub += 1 << k;
            const int sc = lut(MC_SUBCONFIG13_OFF + sub);
            if (sc == 0) MC_PICK(MC_ROW(TILING13_1, cfg), 4);
            else if (sc <= 6) MC_PICK(MC_ROW3(TILING13_2, cfg, sc - 1), 6);
            else if (sc <= 18) MC_PICK(MC_ROW3(TILING13_3, cfg, sc - 7), 10);
            else if (sc <= 22) MC_PICK(MC_ROW3(TILING13_4, cfg, sc - 19), 12);
            else if (sc <= 26) {
                if (test_interior(c, kase, cfg, sc - 23, MC_T2(TEST13, cfg, 6))) MC_PICK(MC_ROW3(TILING13_5_1, cfg, sc - 23), 6);
                else MC_PICK(MC_ROW3(TILING13_5_2, cfg, sc - 23), 10);
            } else if (sc <= 38) MC_PICK(MC_ROW3(TILING13_3_, cfg, sc - 27), 10);
            else if (sc <= 44) MC_PICK(MC_ROW3(TILING13_2_, cfg, sc - 39), 6);
            else if (sc == 45) MC_PICK(MC_ROW(TILING13_1_, cfg), 4);
            break;
        }
        case 14: MC_PICK(MC_ROW(TILING14, cfg), 4); break;
        default: break;
    }
#undef MC_PICK
}

// ---- edge bookkeeping ------------------------------------------------------------------------------
// corner offsets (dz,dy,dx) of the two ends of edge e, Lewiner numbering
__device__ __constant__ signed char MC_EDGE_A[12][3] = {{0,0,0},{0,0,1},{0,1,1},{0,1,0},{1,0,0},{1,0,1},{1,1,1},{1,1,0},{0,0,0},{0,0,1},{0,1,1},{0,1,0}};
__device__ __constant__ signed char MC_EDGE_B[12][3] = {{0,0,1},{0,1,1},{0,1,0},{0,0,0},{1,0,1},{1,1,1},{1,1,0},{1,0,0},{1,0,0},{1,0,1},{1,1,1},{1,1,0}};
// axis of the edge (0 = x / axis2, 1 = y / axis1, 2 = z / axis0) and its lower corner offset (dz,dy,dx)
__device__ __constant__ signed char MC_EDGE_AXIS[12] = {0, 1, 0, 1, 0, 1, 0, 1, 2, 2, 2, 2};
__device__ __constant__ signed char MC_EDGE_LO[12][3] = {{0,0,0},{0,0,1},{0,1,0},{0,0,0},{1,0,0},{1,0,1},{1,1,0},{1,0,0},{0,0,0},{0,0,1},{0,1,1},{0,1,0}};
// Lewiner corner -> skimage's "bitwise" corner index dz*4+dy*2+dx used for vv[] / vg[]
__device__ __forceinline__ int bitwise_index(const signed char* o) { return o[0] * 4 + o[1] * 2 + o[2]; }

// this cube creates the vertex on edge e iff no cube earlier in scan order contains that edge
__device__ __forceinline__ bool owns_edge(int e, int z, int y, int x) {
    switch (e) {
        case 0: return y == 0 && z == 0;
        case 2: return z == 0;
        case 4: return y == 0;
        case 6: return true;
        case 3: return x == 0 && z == 0;
        case 1: return z == 0;
        case 7: return x == 0;
        case 5: return true;
        case 8: return x == 0 && y == 0;
        case 9: return y == 0;
        case 11: return x == 0;
        default: return true;   // 10, and the centre vertex (12)
    }
}

__device__ __forceinline__ int count_created(int offset, int nt, int z, int y, int x) {
    unsigned seen = 0;
    int n = 0;
    for (int i = 0; i < 3 * nt; ++i) {
        const int e = lut(offset + i);
        if (seen & (1u << e)) continue;
        seen |= 1u << e;
        n += owns_edge(e, z, y, x) ? 1 : 0;
    }
    return n;
}

// code byte per cube: low nibble = triangles (<= 12), high nibble = vertices this cube creates (<= 13).
// The tiling row itself is NOT stored: the ~1 % of cubes that are active re-derive it in the emit passes,
// which keeps the per-cube state at 1 B instead of 4 B (the volume itself is 4 B / voxel).
__device__ __forceinline__ uint32_t pack_code(int nt, int ncreated) { return (uint32_t)nt | ((uint32_t)ncreated << 4); }
__device__ __forceinline__ int code_nt(uint32_t c) { return c & 0xf; }
__device__ __forceinline__ int code_created(uint32_t c) { return (c >> 4) & 0xf; }

// (z, y, x) of cube `id`; 32-bit arithmetic whenever the cube count allows (64-bit division is emulated)
__device__ __forceinline__ void cube_coords(const McDims& d, int64_t id, int& z, int& y, int& x) {
    if (d.cubes < (int64_t(1) << 31)) {
        const uint32_t plane = (uint32_t)d.c1 * (uint32_t)d.c2, i = (uint32_t)id;
        const uint32_t zz = i / plane, r = i - zz * plane, yy = r / (uint32_t)d.c2;
        z = (int)zz; y = (int)yy; x = (int)(r - yy * (uint32_t)d.c2);
    } else {
        const int64_t plane = (int64_t)d.c1 * d.c2;
        z = (int)(id / plane);
        const int64_t r = id - (int64_t)z * plane;
        y = (int)(r / d.c2);
        x = (int)(r - (int64_t)y * d.c2);
    }
}

__device__ __forceinline__ void next_cube(const McDims& d, int& z, int& y, int& x) {
    if (++x == d.c2) { x = 0; if (++y == d.c1) { y = 0; ++z; } }
}

// corner sign pattern without forming the differences: (double)v - iso > 0  <=>  (double)v > iso
__device__ __forceinline__ int cube_index(const float* __restrict__ vol, const McDims& d, int z, int y, int x, double iso) {
    const int64_t s1 = d.n2, s0 = (int64_t)d.n1 * d.n2;
    const float* p = vol + (int64_t)z * s0 + (int64_t)y * s1 + x;
    int index = 0;
    index |= ((double)p[0] > iso) ? 1 : 0;            index |= ((double)p[1] > iso) ? 2 : 0;
    index |= ((double)p[s1 + 1] > iso) ? 4 : 0;       index |= ((double)p[s1] > iso) ? 8 : 0;
    index |= ((double)p[s0] > iso) ? 16 : 0;          index |= ((double)p[s0 + 1] > iso) ? 32 : 0;
    index |= ((double)p[s0 + s1 + 1] > iso) ? 64 : 0; index |= ((double)p[s0 + s1] > iso) ? 128 : 0;
    return index;
}

__device__ __forceinline__ int index_of(const Cube& c) {
    int index = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) index |= (c.v[k] > 0.0) ? (1 << k) : 0;
    return index;
}

// ---- pass 1: classify ---------------------------------------------------------------------------------
// Two kernels (round 2; round 1 did both in one 118-VGPR kernel with a 32 KB LDS queue: 4 workgroups per CU, 0.94 TB/s,
// 1.42x over-fetch).
//
// (a) mc_classify_stream -- the HBM-bound part, and nothing else.  A workgroup of 1024 threads = 128 x-groups (512
//     voxels: a whole row of the 480^3 grid) x 8 adjacent rows marches `zrun` planes up axis 0: per plane every thread
//     issues ONE 16-byte load (the workgroup: one contiguous 15 KB run), MC_AHEAD planes in flight.  Voxels fetched
//     per cube: 8/7 x 25/24 = 1.19, all of it in whole rows -- round 1's bricks (129 x 9 x 9 voxels in 516-byte row
//     segments) measured 1.42x whatever the brick shape.  Every load is UNCONDITIONAL (addresses clamped into the
//     volume; clamped values only reach cubes that do not exist): loads inside divergent branches make the compiler
//     fall back to s_waitcnt vmcnt(0) at every use.
//     Per voxel the only work is the sign test: 4 v_cmp per lane and step, whose 64 results per wavefront land in a
//     scalar register pair.  "Which cubes does the surface cut" -- the 8 corner bits are neither all 0 nor all 1 -- is
//     bit-parallel logic on those 64-bit masks, and ONE wavefront does it for the whole workgroup: the 16 waves leave
//     their 4 masks in LDS (32 B each), and after the step's barrier the last wave combines the masks of every
//     wave-row (its own row and the one above, this step and the previous one, the neighbour wave's first column)
//     (lane = wave-row x column, 56 lanes) into cut masks with a dozen 64-bit VALU operations -- one pass.  The
//     owners pick their cut masks up after the NEXT barrier and append the (few) cut cubes lane-parallel to an LDS
//     staging buffer; a per-wave-row flag lets the ~90 % of wavefront rows that do not touch the surface skip that
//     with one scalar branch.
//     History of this kernel at 480^3 (tests/tools/membw.hip streams the same bytes in the same workgroup shape,
//     barrier per step included, in 76 us; this kernel with the logic removed: 97 us): 8-bit pattern assembled per
//     cube on the VALU (~30 ops / cube) 244 us; mask logic on the scalar unit of every wave (~60 s_and / s_or per
//     wave and step: the CU's one scalar ALU becomes the bottleneck) 154 us; one logic wave per workgroup: see
//     DESIGN.md 3.3.
//     The staging buffer goes to the global queue (one atomicAdd) at the end of the march, earlier only if it could
//     overflow.  The code bytes of all other cubes are a memset.
// (b) mc_classify_cut -- one thread per queued cube: corner pattern, the MC33 face / interior tests (fp64), the tiling
//     row, the number of vertices the cube creates; writes that cube's code byte.
constexpr int MC_ZRUN_MIN = 8, MC_ZRUN_MAX = 24;                    // planes per march: chosen per launch (mc_pick_zrun)
constexpr int MC_GX = 128, MC_GY = 8;                               // thread grid: x-groups per row, rows (7 cube rows + halo)
constexpr int MC_STREAM_THREADS = MC_GX * MC_GY;
constexpr int MC_CUBES_PER_PLANE = MC_GX * MC_ITEMS * (MC_GY - 1);  // 3584 cubes per step of the march
constexpr int MC_STAGE = 12288;                                     // LDS staging entries (48 KB)
constexpr int MC_FLUSH_AT = MC_STAGE - 2 * MC_CUBES_PER_PLANE;      // see the flush decision below
constexpr int MC_AHEAD = 4;   // even (LDS double buffers are indexed by the ring slot's parity).  6 needs 67 VGPRs: 3 spills, one
                              // of them reloaded -- behind an s_waitcnt vmcnt(0), i.e. behind all its prefetches -- by the logic wave every step
constexpr bool MC_MARCH0 = true;   // march along axis 0, thread rows = adjacent rows of axis 1 (false: the other way round; same speed)
constexpr int MC_LOGIC_WAVE = MC_STREAM_THREADS / 64 - 1;           // second half of the halo row: owns no cubes
static_assert(MC_GX == 128 && MC_ITEMS == 4 && 8 * (MC_GY - 1) <= 64 && MC_AHEAD % 2 == 0, "two wavefronts per row, four voxels per lane");

struct __attribute__((packed)) McWord { uint32_t v; };              // 4 code bytes at any byte offset

// Sign masks of one row as a wavefront sees it: mask j, bit L = voxel 4 L + j of the wave's 256-voxel span; "column 4"
// (voxel 4 L + 4) is column 0 shifted down one lane with the neighbour wave's / halo voxel's bit on top.
template <bool VEC, bool XHALO>   // VEC: n2 % 4 == 0 (every x-group is one 16-byte load); XHALO: more than one brick in x
__global__ __launch_bounds__(MC_STREAM_THREADS, 8) void mc_classify_stream(const float* __restrict__ vol, McDims d, float thr,
                                                                        uint64_t* __restrict__ queue,
                                                                        unsigned long long* __restrict__ qcount, const int zrun) {
    __shared__ uint64_t s_m[2][MC_GY][2][4];             // sign masks   [step parity][row][x half][column]
    __shared__ uint64_t s_cut[2][MC_GY - 1][2][4];       // cut masks    [step parity][cube row][x half][column]
    __shared__ uint32_t s_some[2][MC_GY - 1][2];         // any bit set in the 4 cut masks
    __shared__ uint64_t s_prev[2 * (MC_GY - 1)][2][5];   // logic wave: [wave-row][AND, OR][column] of the previous step
    __shared__ uint32_t s_h[2][MC_GY];                   // the voxel behind the brick
    __shared__ uint32_t s_q[MC_STAGE];
    __shared__ uint32_t s_n, s_snap[2];
    __shared__ unsigned long long s_base;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int xh = wave & 1, tz = wave >> 1, tx = xh * 64 + lane;
    // thread rows along axis R (origin r0), the march along axis M (origin m0): (R, M) = (1, 0) or (0, 1)
    const int xb = blockIdx.x * MC_GX * MC_ITEMS, r0 = blockIdx.y * (MC_GY - 1), m0 = blockIdx.z * zrun;
    const int nr = MC_MARCH0 ? d.n1 : d.n0, nm = MC_MARCH0 ? d.n0 : d.n1, cr = MC_MARCH0 ? d.c1 : d.c0, cm = MC_MARCH0 ? d.c0 : d.c1;
    const int64_t sr = MC_MARCH0 ? (int64_t)d.n2 : (int64_t)d.n1 * d.n2, sm = MC_MARCH0 ? (int64_t)d.n1 * d.n2 : (int64_t)d.n2;
    const int x0 = xb + tx * MC_ITEMS, r = r0 + tz;
    const int xl = min(x0, VEC ? d.n2 - 4 : d.n2 - 1);
    const float* rowp = vol + (int64_t)min(r, nr - 1) * sr;
    if (threadIdx.x == 0) { s_n = 0; s_snap[0] = s_snap[1] = 0; }
    struct __attribute__((packed, aligned(4))) F4 { float v[4]; };
    auto fetch = [&](int l) -> F4 {
        const float* p = rowp + (int64_t)min(m0 + l, nm - 1) * sm;
        F4 q;
        if constexpr (VEC) q = *reinterpret_cast<const F4*>(p + xl);
        else {
#pragma unroll
            for (int k = 0; k < 4; ++k) q.v[k] = p[min(xl + k, d.n2 - 1)];
        }
        return q;
    };
    auto fetch_halo = [&](int l) -> float {   // the voxel behind the brick (asked for by the last x-group only; everybody loads)
        const float* p = rowp + (int64_t)min(m0 + l, nm - 1) * sm;
        return p[tx == MC_GX - 1 ? min(xb + MC_GX * MC_ITEMS, d.n2 - 1) : min(xl, d.n2 - 1)];
    };
    F4 pf[MC_AHEAD];
    float ph[XHALO ? MC_AHEAD : 1];
#pragma unroll
    for (int l = 0; l < MC_AHEAD; ++l) {
        pf[l] = fetch(l);
        if constexpr (XHALO) ph[l] = fetch_halo(l);
    }
    auto flush = [&]() {                                 // entered by every wave, or by none
        if (threadIdx.x == 0) s_base = atomicAdd(qcount, (unsigned long long)s_n);
        __syncthreads();
        const uint32_t cnt = s_n;
        for (uint32_t e = threadIdx.x; e < cnt; e += MC_STREAM_THREADS) {
            const uint32_t local = s_q[e];
            const int k = local % (MC_GX * MC_ITEMS), rr = (local / (MC_GX * MC_ITEMS)) % (MC_GY - 1), ll = local / MC_CUBES_PER_PLANE;
            const int cz = MC_MARCH0 ? m0 + ll : r0 + rr, cy = MC_MARCH0 ? r0 + rr : m0 + ll;
            queue[s_base + e] = (uint64_t)(((int64_t)cz * d.c1 + cy) * d.c2 + xb + k);
        }
        __syncthreads();
        if (threadIdx.x == 0) s_n = 0;
        __syncthreads();
    };
    // owners: append the cut cubes of march step `lc` (cubes between voxel planes lc and lc + 1) from the cut masks
    auto append = [&](int lc) {
        if (tz >= MC_GY - 1) return;
        const int par = (lc + 1) & 1;                    // written by the logic wave in iteration lc + 1
        if (__builtin_amdgcn_readfirstlane((int)s_some[par][tz][xh]) == 0) return;   // scalar branch: nothing cut here
        const uint32_t base = (uint32_t)(lc * MC_CUBES_PER_PLANE + tz * (MC_GX * MC_ITEMS) + tx * MC_ITEMS);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (((s_cut[par][tz][xh][j] >> lane) & 1u) && x0 + j < d.c2) s_q[atomicAdd(&s_n, 1u)] = base + j;
    };
    // logic wave: lane = 4 k + column, k = 2 cube row + x half (56 lanes)
    const int kk = lane >> 2, kcol = lane & 3, krow = min(kk >> 1, MC_GY - 2), khalf = kk & 1;
    const bool k_exists = r0 + krow < cr;
    // (that state -- 10 masks per lane -- lives in LDS, not in registers every wave would have to reserve: at 64 VGPRs
    // two workgroups share a CU)
    __syncthreads();
    // the march: an outer loop over groups of MC_AHEAD planes (not unrolled: the fully unrolled loop made the register
    // allocator spill the planes in flight at 64 VGPRs), the ring slot of a plane is its position in the group
#pragma unroll 1
    for (int l0 = 0; l0 <= zrun; l0 += MC_AHEAD) {
#pragma unroll
      for (int u = 0; u < MC_AHEAD; ++u) {
        const int l = l0 + u;                            // voxel plane m0 + l
        if (l > zrun) break;
        const F4 q = pf[u];
        float hq = 0.0f;
        if constexpr (XHALO) hq = ph[u];
        if (l + MC_AHEAD <= zrun) {                      // MC_AHEAD planes in flight
            pf[u] = fetch(l + MC_AHEAD);
            if constexpr (XHALO) ph[u] = fetch_halo(l + MC_AHEAD);
        }
        uint64_t b[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = __ballot(q.v[j] > thr);
        if (lane == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) s_m[u & 1][tz][xh][j] = b[j];
        }
        if constexpr (XHALO) {
            const uint64_t hm = __ballot(hq > thr);
            if (xh == 1 && lane == 0) s_h[u & 1][tz] = (uint32_t)(hm >> 63);
        }
        // Flush decision, identical in every wave: thread 0's snapshot of the fill level, taken BEFORE this barrier
        // (it counts every append up to two iterations back, possibly some of the previous one's), read after it.
        // Without a flush the buffer holds at most snapshot + 2 steps' worth of appends at the next decision.
        if (threadIdx.x == 0) s_snap[u & 1] = s_n;
        __syncthreads();
        if ((uint32_t)__builtin_amdgcn_readfirstlane((int)s_snap[u & 1]) > (uint32_t)MC_FLUSH_AT) flush();
        if (l >= 2) append(l - 2);                       // masks the logic wave wrote during the previous iteration
        if (wave == MC_LOGIC_WAVE) {
            // lane (k, j), k = wave-row (krow, khalf), j = column: the cut mask of the 64 cubes 4 L + j of that wave-row
            // from columns j and j + 1 of two rows (krow, krow + 1) and two planes (the previous step's AND / OR per
            // column are kept in s_prev; every lane reads before any lane writes: one wavefront, LDS in program order)
            const int par = u & 1;
            uint64_t cut = 0;
            if (lane < 8 * (MC_GY - 1)) {
                const int jn = (kcol + 1) & 3;
                const uint64_t own = s_m[par][krow][khalf][kcol], up = s_m[par][krow + 1][khalf][kcol];
                const uint64_t own_n = s_m[par][krow][khalf][jn], up_n = s_m[par][krow + 1][khalf][jn];
                const uint64_t and_c = own & up, or_c = own | up;
                uint64_t and_n = own_n & up_n, or_n = own_n | up_n;
                if (kcol == 3) {   // column 4 = column 0 one lane up; the top bit: next wave of the row / the voxel behind the brick
                    uint64_t nb_own, nb_up;
                    if (khalf == 0) { nb_own = s_m[par][krow][1][0] & 1u; nb_up = s_m[par][krow + 1][1][0] & 1u; }
                    else { nb_own = XHALO ? s_h[par][krow] : 0u; nb_up = XHALO ? s_h[par][krow + 1] : 0u; }
                    and_n = (and_n >> 1) | ((nb_own & nb_up) << 63);
                    or_n = (or_n >> 1) | ((nb_own | nb_up) << 63);
                }
                const uint64_t all1 = s_prev[kk][0][kcol] & and_c & s_prev[kk][0][kcol + 1] & and_n;
                const uint64_t any1 = s_prev[kk][1][kcol] | or_c | s_prev[kk][1][kcol + 1] | or_n;
                if (l > 0 && k_exists && m0 + l - 1 < cm) cut = any1 & ~all1;      // cubes of step l - 1
                s_prev[kk][0][kcol] = and_c;
                s_prev[kk][1][kcol] = or_c;
                if (kcol == 3) { s_prev[kk][0][4] = and_n; s_prev[kk][1][4] = or_n; }
                s_cut[par][krow][khalf][kcol] = cut;     // (lanes past the end of a row hold copies of its last voxels; append() masks them)
            }
            const uint64_t nonzero = __ballot(cut != 0);
            if (lane < 2 * (MC_GY - 1)) s_some[par][lane >> 1][lane & 1] = (uint32_t)((nonzero >> (4 * lane)) & 15u);
        }
      }
    }
    __syncthreads();                                     // the logic wave's last masks; every append so far is counted in s_n
    // The in-loop decision bounds the fill at MC_STAGE after the last iteration's appends (snapshot <= MC_FLUSH_AT, plus
    // at most one step the snapshot missed, plus that iteration's step) -- the final step below needs its own room.
    // s_n is exact and stable here (nobody appends between the barrier above and the one inside flush()).
    if ((uint32_t)__builtin_amdgcn_readfirstlane((int)s_n) > (uint32_t)(MC_STAGE - MC_CUBES_PER_PLANE)) flush();
    append(zrun - 1);
    __syncthreads();
    flush();
}
static_assert(MC_FLUSH_AT > 0 && (int64_t)MC_ZRUN_MAX * MC_CUBES_PER_PLANE < (int64_t(1) << 32), "staging buffer / local id");

// Planes per march: 24 (halo planes cost 25/24; 1380 workgroups at 480^3), fewer for volumes that would not give every
// CU its two workgroups.  (Choosing zrun to make the launch a whole number of "rounds" -- 22 at 480^3 -- measures
// slower, 124.5 vs 121 us: workgroups do not run in lockstep rounds, the extra halo planes are what counts.)
static int mc_pick_zrun(int64_t bricks_xy, int c_march, int num_cus) {
    const int64_t slots = 2 * (int64_t)(num_cus > 0 ? num_cus : 256);
    int z = 24;
    while (z > MC_ZRUN_MIN && bricks_xy * ((c_march + z - 1) / z) < slots) --z;
    return z;
}

__global__ __launch_bounds__(MC_BLOCK) void mc_classify_cut(const float* __restrict__ vol, McDims d, double iso,
                                                            const uint64_t* __restrict__ queue,
                                                            const unsigned long long* __restrict__ qcount,
                                                            uint8_t* __restrict__ codes, uint8_t* __restrict__ tile_flag) {
    const int64_t n = (int64_t)*qcount;
    for (int64_t q = (int64_t)blockIdx.x * MC_BLOCK + threadIdx.x; q < n; q += (int64_t)gridDim.x * MC_BLOCK) {
        const int64_t id = (int64_t)queue[q];
        int z, y, x;
        cube_coords(d, id, z, y, x);
        Cube c;
        load_cube(vol, d, z, y, x, iso, c);
        int off, nt;
        select_tiling(c, index_of(c), off, nt);
        if (nt > 0) {
            codes[id] = (uint8_t)pack_code(nt, count_created(off, nt, z, y, x));
            // mark the cube's tile: the passes over the code bytes skip unmarked tiles (~60 % at 480^3).  (Adding the
            // tile sums here with integer atomics instead costs more than the pass it saves: 35 -> 133 us, neighbouring
            // queue entries hit the same counters.)
            tile_flag[id / MC_TILE] = 1;
        }
    }
}

// the 16 code bytes of cubes first .. first + 15 (bytes past the last cube read as 0)
__device__ __forceinline__ uint4 load_codes16(const McDims& d, const uint32_t* __restrict__ codes4, int64_t first) {
    if (first >= d.cubes) return make_uint4(0, 0, 0, 0);
    uint4 w = *reinterpret_cast<const uint4*>(codes4 + (first >> 2));      // the buffer is padded by 4 KiB
    const int64_t left = d.cubes - first;                                  // cubes from `first` to the end
    uint32_t* p = &w.x;
    for (int q = 0; q < 4; ++q) {
        const int64_t have = left - 4 * q;
        if (have <= 0) p[q] = 0u;
        else if (have < 4) p[q] &= (1u << (8 * (int)have)) - 1u;
    }
    return w;
}

// per-tile (1024 cubes in scan order) sums of created vertices / triangles / active cubes, from the code bytes:
// one wavefront per tile, 16 cubes (one 16-byte load) per lane (a workgroup per tile is dispatch-bound: 91 us); tiles
// mc_classify_cut did not mark are all zero and are not read
__global__ __launch_bounds__(256) void mc_tile_sums(McDims d, const uint32_t* __restrict__ codes4, int64_t tiles,
                                                    const uint8_t* __restrict__ tile_flag, uint4* __restrict__ tile_sums) {
    const int lane = threadIdx.x & 63;
    const int64_t tile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= tiles) return;
    if (!tile_flag[tile]) {
        if (lane == 0) tile_sums[tile] = make_uint4(0, 0, 0, 0);
        return;
    }
    const uint4 w = load_codes16(d, codes4, tile * MC_TILE + lane * 16);
    uint32_t nv = 0, ntri = 0, nact = 0;
    const uint32_t* p = &w.x;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const uint32_t c = p[q] >> (8 * it);
            nv += code_created(c); ntri += code_nt(c); nact += code_nt(c) ? 1 : 0;
        }
    for (int o = 32; o > 0; o >>= 1) { nv += __shfl_xor(nv, o); ntri += __shfl_xor(ntri, o); nact += __shfl_xor(nact, o); }
    if (lane == 0) tile_sums[tile] = make_uint4(nv, ntri, nact, 0);
}

// ---- pass 2: exclusive scan of the per-tile sums: 1024-tile groups in parallel, then the group totals ------
__device__ __forceinline__ void block_scan3(uint32_t (&inc)[3], uint32_t (&tot)[3], uint32_t (*s_w)[16]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;   // 1024 threads: inclusive scan in place
#pragma unroll
    for (int k = 0; k < 3; ++k)
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t pv = __shfl_up(inc[k], o);
            if (lane >= o) inc[k] += pv;
        }
    if (lane == 63) { s_w[0][wave] = inc[0]; s_w[1][wave] = inc[1]; s_w[2][wave] = inc[2]; }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        uint32_t w = lane < 16 ? s_w[k][lane] : 0u;
        for (int o = 1; o < 16; o <<= 1) {
            const uint32_t pw = __shfl_up(w, o);
            if (lane >= o) w += pw;
        }
        tot[k] = __shfl(w, 15);
        const uint32_t below = __shfl(w, wave > 0 ? wave - 1 : 0);
        inc[k] += wave > 0 ? below : 0u;
    }
}

__global__ __launch_bounds__(1024) void mc_scan_groups(uint4* __restrict__ tile_sums, int64_t tiles,
                                                       uint4* __restrict__ group_sums) {
    __shared__ uint32_t s_w[3][16];
    const int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x;
    const uint4 v = i < tiles ? tile_sums[i] : make_uint4(0, 0, 0, 0);
    uint32_t inc[3] = {v.x, v.y, v.z}, tot[3];
    block_scan3(inc, tot, s_w);
    if (i < tiles) tile_sums[i] = make_uint4(inc[0] - v.x, inc[1] - v.y, inc[2] - v.z, v.z);  // exclusive within the group; .w = the tile's own active cubes
    if (threadIdx.x == 0) group_sums[blockIdx.x] = make_uint4(tot[0], tot[1], tot[2], 0);
}

__global__ __launch_bounds__(1024) void mc_scan_totals(uint4* __restrict__ group_sums, int64_t groups,
                                                       uint32_t* __restrict__ totals) {
    __shared__ uint32_t s_w[3][16];
    uint32_t carry[3] = {0, 0, 0};
    for (int64_t start = 0; start < groups; start += 1024) {      // one round up to 1M tiles = 1G cubes
        const int64_t i = start + threadIdx.x;
        const uint4 v = i < groups ? group_sums[i] : make_uint4(0, 0, 0, 0);
        uint32_t inc[3] = {v.x, v.y, v.z}, tot[3];
        block_scan3(inc, tot, s_w);
        if (i < groups) group_sums[i] = make_uint4(carry[0] + inc[0] - v.x, carry[1] + inc[1] - v.y, carry[2] + inc[2] - v.z, 0);
#pragma unroll
        for (int k = 0; k < 3; ++k) carry[k] += tot[k];
        __syncthreads();
    }
    if (threadIdx.x == 0) { totals[0] = carry[0]; totals[1] = carry[1]; totals[2] = carry[2]; }
}

struct McOut {
    float* verts;      // (V,3) in (axis0, axis1, axis2) order
    int32_t* faces;    // (F,3)
    float* normals;    // (V,3)
    float* values;     // (V,)
    int32_t* edge_vertex[3];   // per axis (x,y,z): vertex id of the edge whose lower corner is the voxel
};

struct McActive {      // one entry per cube that emits triangles, in scan order
    int64_t id;
    uint32_t vbase, tbase;
};

// ---- pass 3a: compact the active cubes (scan order) with their vertex / triangle bases -----------------------
// one wavefront per tile, 16 cubes per lane, wave-level exclusive scan
__global__ __launch_bounds__(256) void mc_compact(McDims d, const uint32_t* __restrict__ codes4, int64_t tiles,
                                                  const uint4* __restrict__ tile_prefix,
                                                  const uint4* __restrict__ group_prefix, McActive* __restrict__ list) {
    const int lane = threadIdx.x & 63;
    const int64_t tile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= tiles) return;
    const uint4 tp = tile_prefix[tile];
    if (tp.w == 0) return;                                 // nothing cut in this tile (~60 % of them): its code bytes are not read
    const int64_t first = tile * MC_TILE + lane * 16;
    const uint4 w = load_codes16(d, codes4, first);
    const uint32_t* p = &w.x;
    uint32_t own[3] = {0, 0, 0};
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const uint32_t c = p[q] >> (8 * it);
            own[0] += code_created(c); own[1] += code_nt(c); own[2] += code_nt(c) ? 1 : 0;
        }
    uint32_t inc[3] = {own[0], own[1], own[2]};
#pragma unroll
    for (int k = 0; k < 3; ++k)
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t pv = __shfl_up(inc[k], o);
            if (lane >= o) inc[k] += pv;
        }
    if (own[2] == 0) return;
    const uint4 gp = group_prefix[tile >> 10];
    uint32_t pre[3] = {gp.x + tp.x + inc[0] - own[0], gp.y + tp.y + inc[1] - own[1], gp.z + tp.z + inc[2] - own[2]};
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const uint32_t c = p[q] >> (8 * it);
            if (code_nt(c)) {
                McActive e; e.id = first + 4 * q + it; e.vbase = pre[0]; e.tbase = pre[1];
                list[pre[2]++] = e;
            }
            pre[0] += code_created(c); pre[1] += code_nt(c);
        }
}

// ---- pass 3b: vertices and faces, one thread per ACTIVE cube (all lanes busy) ---------------------------------
template <bool FACES>
__global__ __launch_bounds__(256) void mc_emit(const float* __restrict__ vol, McDims d, double iso,
                                               const McActive* __restrict__ list, const uint32_t* __restrict__ totals,
                                               McOut out, int64_t* __restrict__ vertex_cube,
                                               int8_t* __restrict__ vertex_edge) {
    const int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= (int64_t)totals[2]) return;     // number of active cubes, left on the device by the scan
    const McActive ent = list[a];
    int z, y, x;
    cube_coords(d, ent.id, z, y, x);
    Cube c;
    load_cube(vol, d, z, y, x, iso, c);
    int off, nt;
    select_tiling(c, index_of(c), off, nt);
    if constexpr (!FACES) {
        unsigned seen = 0;
        uint32_t next = ent.vbase;
        for (int i = 0; i < 3 * nt; ++i) {
            const int e = lut(off + i);
            if (seen & (1u << e)) continue;
            seen |= 1u << e;
            if (!owns_edge(e, z, y, x)) continue;
            double px, py, pz;
            if (e == 12) {
                double fx = 0, fy = 0, fz = 0, ff = 0;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const double w = 1.0 / (SK_EPS + fabs(c.v[k]));
                    const int cx = (k == 1 || k == 2 || k == 5 || k == 6), cy = (k == 2 || k == 3 || k == 6 || k == 7), cz = k >> 2;
                    fx += (double)cx * w; fy += (double)cy * w; fz += (double)cz * w; ff += w;
                }
                px = x + fx / ff; py = y + fy / ff; pz = z + fz / ff;
            } else {
                const signed char* ea = MC_EDGE_A[e];
                const signed char* eb = MC_EDGE_B[e];
                const int ka = e < 8 ? ((e & 3)) + (e & 4) : e - 8;           // Lewiner corner of end A
                const int kb = e < 8 ? (((e & 3) + 1) & 3) + (e & 4) : e - 4; // Lewiner corner of end B
                const double w1 = 1.0 / (SK_EPS + fabs(pick(c, ka))), w2 = 1.0 / (SK_EPS + fabs(pick(c, kb)));
                double fx = 0, fy = 0, fz = 0, ff = 0;
                fx += (double)ea[2] * w1; fy += (double)ea[1] * w1; fz += (double)ea[0] * w1; ff += w1;
                fx += (double)eb[2] * w2; fy += (double)eb[1] * w2; fz += (double)eb[0] * w2; ff += w2;
                px = x + fx / ff; py = y + fy / ff; pz = z + fz / ff;
                const signed char* lo = MC_EDGE_LO[e];
                const int64_t vox = ((int64_t)(z + lo[0]) * d.n1 + (y + lo[1])) * d.n2 + (x + lo[2]);
                out.edge_vertex[MC_EDGE_AXIS[e]][vox] = (int32_t)next;
            }
            // wrapper: vertices flipped to (axis0, axis1, axis2) = (z, y, x)
            out.verts[3 * (int64_t)next] = (float)pz;
            out.verts[3 * (int64_t)next + 1] = (float)py;
            out.verts[3 * (int64_t)next + 2] = (float)px;
            vertex_cube[next] = ent.id;
            vertex_edge[next] = (int8_t)e;
            ++next;
        }
    } else {
        // the centre vertex (if any) is created by this cube: its id = vbase + #owned first-uses before it
        int centre = -1;
        {
            unsigned seen = 0;
            int k = 0;
            for (int i = 0; i < 3 * nt; ++i) {
                const int e = lut(off + i);
                if (seen & (1u << e)) continue;
                seen |= 1u << e;
                if (e == 12) { centre = (int)ent.vbase + k; break; }
                k += owns_edge(e, z, y, x) ? 1 : 0;
            }
        }
        for (int t = 0; t < nt; ++t) {
            int idx[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int e = lut(off + 3 * t + j);
                if (e == 12) { idx[j] = centre; continue; }
                const signed char* lo = MC_EDGE_LO[e];
                const int64_t vox = ((int64_t)(z + lo[0]) * d.n1 + (y + lo[1])) * d.n2 + (x + lo[2]);
                idx[j] = out.edge_vertex[MC_EDGE_AXIS[e]][vox];
            }
            int32_t* f = out.faces + 3 * ((int64_t)ent.tbase + t);
            f[0] = idx[2]; f[1] = idx[1]; f[2] = idx[0];   // gradient_direction='descent' reverses each triple
        }
    }
}

// ---- pass 5: normals + values (one thread per vertex, replaying its cubes in scan order) ---------------------
__device__ __forceinline__ void corner_gradients(const Cube& c, double (&g)[24]) {
    const double* v = c.v;
    g[0] = v[0] - v[1];  g[1] = v[0] - v[3];  g[2] = v[0] - v[4];
    g[3] = v[0] - v[1];  g[4] = v[1] - v[2];  g[5] = v[1] - v[5];
    g[6] = v[3] - v[2];  g[7] = v[1] - v[2];  g[8] = v[2] - v[6];
    g[9] = v[3] - v[2];  g[10] = v[0] - v[3]; g[11] = v[3] - v[7];
    g[12] = v[4] - v[5]; g[13] = v[4] - v[7]; g[14] = v[0] - v[4];
    g[15] = v[4] - v[5]; g[16] = v[5] - v[6]; g[17] = v[1] - v[5];
    g[18] = v[7] - v[6]; g[19] = v[5] - v[6]; g[20] = v[2] - v[6];
    g[21] = v[7] - v[6]; g[22] = v[4] - v[7]; g[23] = v[3] - v[7];
}

// the cubes sharing an edge, as offsets (dz,dy,dx) from the creating... from the edge's LOWER CORNER voxel,
// in scan order, with the edge's local id inside each: [axis][k] -> {dz, dy, dx, local edge}
__device__ __constant__ signed char MC_SHARE[3][4][4] = {
    {{-1, -1, 0, 6}, {-1, 0, 0, 4}, {0, -1, 0, 2}, {0, 0, 0, 0}},     // x edge
    {{-1, 0, -1, 5}, {-1, 0, 0, 7}, {0, 0, -1, 1}, {0, 0, 0, 3}},     // y edge
    {{0, -1, -1, 10}, {0, -1, 0, 11}, {0, 0, -1, 9}, {0, 0, 0, 8}}};  // z edge

__global__ __launch_bounds__(256) void mc_vertex_attributes(const float* __restrict__ vol, McDims d, double iso,
                                                            const int64_t* __restrict__ vertex_cube,
                                                            const int8_t* __restrict__ vertex_edge, int64_t nverts,
                                                            McOut out) {
    const int64_t vid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (vid >= nverts) return;
    const int64_t home = vertex_cube[vid];
    const int e_home = vertex_edge[vid];
    int hz, hy, hx;
    cube_coords(d, home, hz, hy, hx);
    float nx = 0.0f, ny = 0.0f, nz = 0.0f, value = 0.0f;
    int ncubes = 1, axis = 0, lz = 0, ly = 0, lx = 0;
    if (e_home != 12) {
        axis = MC_EDGE_AXIS[e_home];
        lz = hz + MC_EDGE_LO[e_home][0]; ly = hy + MC_EDGE_LO[e_home][1]; lx = hx + MC_EDGE_LO[e_home][2];
        ncubes = 4;
    }
    for (int k = 0; k < ncubes; ++k) {
        int z, y, x, e;
        if (e_home == 12) { z = hz; y = hy; x = hx; e = 12; }
        else {
            const signed char* s = MC_SHARE[axis][k];
            z = lz + s[0]; y = ly + s[1]; x = lx + s[2]; e = s[3];
            if (z < 0 || y < 0 || x < 0 || z >= d.c0 || y >= d.c1 || x >= d.c2) continue;
        }
        Cube c;
        load_cube(vol, d, z, y, x, iso, c);
        int off, nt;
        select_tiling(c, index_of(c), off, nt);
        if (nt == 0) continue;
        // vmax of the cube = max(v,0) - min(v,0)
        double lo = 0.0, hi = 0.0;
#pragma unroll
        for (int q = 0; q < 8; ++q) { hi = c.v[q] > hi ? c.v[q] : hi; lo = c.v[q] < lo ? c.v[q] : lo; }
        const double vmax = hi - lo;
        double g[24];
        corner_gradients(c, g);
        float gx = 0, gy = 0, gz = 0;   // contribution per reference
        float s1 = 0, s2 = 0;
        double ga[3] = {0, 0, 0}, gb[3] = {0, 0, 0};   // gradients at the two ends of the edge (select chains, no scratch)
        if (e == 12) {
            double w[8], sx = 0, sy = 0, sz = 0;
#pragma unroll
            for (int q = 0; q < 8; ++q) w[q] = 1.0 / (SK_EPS + fabs(c.v[q]));
#pragma unroll
            for (int q = 0; q < 8; ++q) { sx += w[q] * g[3 * q]; sy += w[q] * g[3 * q + 1]; sz += w[q] * g[3 * q + 2]; }
            (void)sx;
            gx = (float)sz; gy = (float)sy; gz = 0.0f;   // skimage's centre gradient: (sum w*gz, sum w*gy, 0)
        } else {
            const int ka = e < 8 ? ((e & 3)) + (e & 4) : e - 8;
            const int kb = e < 8 ? (((e & 3) + 1) & 3) + (e & 4) : e - 4;
            const int i1 = bitwise_index(MC_EDGE_A[e]), i2 = bitwise_index(MC_EDGE_B[e]);
#pragma unroll
            for (int q = 0; q < 8; ++q)
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    ga[a] = i1 == q ? g[3 * q + a] : ga[a];
                    gb[a] = i2 == q ? g[3 * q + a] : gb[a];
                }
            s1 = (float)(1.0 / (SK_EPS + fabs(pick(c, ka))));   // `strength` is a C float in skimage
            s2 = (float)(1.0 / (SK_EPS + fabs(pick(c, kb))));
        }
        bool referenced = false;
        for (int i = 0; i < 3 * nt; ++i) {
            if (lut(off + i) != e) continue;
            referenced = true;
            if (e == 12) { nx += gx; ny += gy; nz += gz; }
            else {
                nx += (float)(ga[0] * (double)s1); ny += (float)(ga[1] * (double)s1); nz += (float)(ga[2] * (double)s1);
                nx += (float)(gb[0] * (double)s2); ny += (float)(gb[1] * (double)s2); nz += (float)(gb[2] * (double)s2);
            }
        }
        if (referenced && vmax > (double)value) value = (float)vmax;
    }
    const double len = sqrt((double)nx * nx + (double)ny * ny + (double)nz * nz);
    if (len > 0.0) { nx = (float)(nx / len); ny = (float)(ny / len); nz = (float)(nz / len); }
    out.normals[3 * vid] = nz; out.normals[3 * vid + 1] = ny; out.normals[3 * vid + 2] = nx;   // flipped columns
    out.values[vid] = value;
}

static inline size_t al(size_t x) { return (x + 255) & ~size_t(255); }

struct McWorkspace {
    uint32_t* codes; uint4* tile_sums; uint8_t* tile_flag; uint4* group_sums; uint32_t* totals; int32_t* edge[3]; int64_t* vertex_cube; int8_t* vertex_edge;
    McActive* active;
    int64_t tiles;
};

static McDims make_dims(int n0, int n1, int n2) {
    McDims d{n0, n1, n2, n0 - 1, n1 - 1, n2 - 1, (int64_t)(n0 - 1) * (n1 - 1) * (n2 - 1)};
    return d;
}

static size_t carve(const McDims& d, char* base, McWorkspace* ws) {
    const int64_t tiles = (d.cubes + MC_TILE - 1) / MC_TILE;
    const size_t vox = (size_t)d.n0 * d.n1 * d.n2;
    size_t off = 0;
    auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += al(bytes); return p; };
    char* p;
    p = take((((size_t)d.cubes + 3) & ~size_t(3)) + 4096); if (ws) ws->codes = (uint32_t*)p;
    p = take((size_t)tiles * 16); if (ws) ws->tile_sums = (uint4*)p;
    p = take((size_t)tiles); if (ws) ws->tile_flag = (uint8_t*)p;
    p = take((size_t)((tiles + 1023) / 1024) * 16); if (ws) ws->group_sums = (uint4*)p;
    p = take(256); if (ws) ws->totals = (uint32_t*)p;
    for (int a = 0; a < 3; ++a) { p = take(vox * 4); if (ws) ws->edge[a] = (int32_t*)p; }
    if (ws) ws->tiles = tiles;
    return off;
}

}  // namespace nm

using namespace nm;

extern "C" {

int64_t nm_mc_workspace_bytes(int32_t n0, int32_t n1, int32_t n2) {
    if (n0 < 2 || n1 < 2 || n2 < 2) return 0;
    return (int64_t)carve(make_dims(n0, n1, n2), nullptr, nullptr);
}

int64_t nm_mc_vertex_scratch_bytes(int64_t vertices, int64_t faces) {
    // per vertex: creating cube (8 B) + edge id (1 B); per active cube (<= faces): one 16-byte list entry
    return (int64_t)(al((size_t)vertices * 8) + al((size_t)vertices) + al((size_t)faces * sizeof(McActive)));
}

int nm_mc_count(const float* d_volume, int32_t n0, int32_t n1, int32_t n2, double iso, void* d_workspace,
                int64_t* h_vertices, int64_t* h_faces, void* stream_) {
    NM_REQUIRE(d_volume && d_workspace && h_vertices && h_faces, "bad argument");
    NM_REQUIRE(n0 >= 2 && n1 >= 2 && n2 >= 2, "Input array must be at least 2x2x2.");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const McDims d = make_dims(n0, n1, n2);
    NM_REQUIRE(d.cubes < (int64_t(1) << 40), "volume too large");
    McWorkspace ws;
    carve(d, static_cast<char*>(d_workspace), &ws);
    const int c_rows = MC_MARCH0 ? d.c1 : d.c0, c_march = MC_MARCH0 ? d.c0 : d.c1;
    static int cus_of_device[64] = {};
    int dev = 0;
    NM_HIP_CHECK(hipGetDevice(&dev));
    if (dev >= 0 && dev < 64 && cus_of_device[dev] == 0) {
        hipDeviceProp_t prop;
        NM_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
        cus_of_device[dev] = prop.multiProcessorCount;
    }
    const unsigned bx = (unsigned)((d.c2 + MC_GX * MC_ITEMS - 1) / (MC_GX * MC_ITEMS)), by = (unsigned)((c_rows + MC_GY - 2) / (MC_GY - 1));
    const int zrun = mc_pick_zrun((int64_t)bx * by, c_march, dev >= 0 && dev < 64 ? cus_of_device[dev] : 0);
    const dim3 bricks(bx, by, (unsigned)((c_march + zrun - 1) / zrun));
    NM_REQUIRE(bricks.y <= 65535u && bricks.z <= 65535u, "volume too large");
    // code bytes: zero everywhere (incl. the <= 3 bytes behind the last cube that share a dword with real cubes);
    // the cut cubes' bytes are written by mc_classify_cut
    NM_HIP_CHECK(hipMemsetAsync(ws.codes, 0, (((size_t)d.cubes + 3) & ~size_t(3)) + 8, stream));
    NM_HIP_CHECK(hipMemsetAsync(ws.totals, 0, 256, stream));
    NM_HIP_CHECK(hipMemsetAsync(ws.tile_flag, 0, (size_t)ws.tiles, stream));        // set by mc_classify_cut
    float thr = (float)iso;                        // largest float <= iso (exact equivalence of the sign test)
    if ((double)thr > iso) thr = nextafterf(thr, -INFINITY);
    // the queue of cut cubes (worst case: every cube) lives in the first two edge->vertex volumes, which nothing
    // touches before nm_mc_emit
    uint64_t* queue = reinterpret_cast<uint64_t*>(ws.edge[0]);
    unsigned long long* qcount = reinterpret_cast<unsigned long long*>(ws.totals + 8);
    const bool vec = n2 % 4 == 0, xhalo = bricks.x > 1;
#define NM_STREAM(V, X) hipLaunchKernelGGL((mc_classify_stream<V, X>), bricks, dim3(MC_STREAM_THREADS), 0, stream, d_volume, d, thr, queue, qcount, zrun)
    if (vec && !xhalo) NM_STREAM(true, false);
    else if (vec) NM_STREAM(true, true);
    else if (!xhalo) NM_STREAM(false, false);
    else NM_STREAM(false, true);
#undef NM_STREAM
    hipLaunchKernelGGL(mc_classify_cut, dim3(2048), dim3(MC_BLOCK), 0, stream, d_volume, d, iso, queue, qcount,
                       reinterpret_cast<uint8_t*>(ws.codes), ws.tile_flag);
    hipLaunchKernelGGL(mc_tile_sums, dim3((unsigned)((ws.tiles + 3) / 4)), dim3(256), 0, stream, d, ws.codes, ws.tiles,
                       ws.tile_flag, ws.tile_sums);
    const int64_t groups = (ws.tiles + 1023) / 1024;
    hipLaunchKernelGGL(mc_scan_groups, dim3((unsigned)groups), dim3(1024), 0, stream, ws.tile_sums, ws.tiles, ws.group_sums);
    hipLaunchKernelGGL(mc_scan_totals, dim3(1), dim3(1024), 0, stream, ws.group_sums, groups, ws.totals);
    NM_HIP_CHECK(hipGetLastError());
    uint32_t totals[3];
    NM_HIP_CHECK(hipMemcpyAsync(totals, ws.totals, sizeof(totals), hipMemcpyDeviceToHost, stream));
    NM_HIP_CHECK(hipStreamSynchronize(stream));
    *h_vertices = totals[0];
    *h_faces = totals[1];
    return 0;
}

int nm_mc_emit(const float* d_volume, int32_t n0, int32_t n1, int32_t n2, double iso, void* d_workspace,
               void* d_vertex_scratch, int64_t vertices, int64_t faces, float* d_verts, int32_t* d_faces,
               float* d_normals, float* d_values, void* stream_) {
    NM_REQUIRE(d_volume && d_workspace && d_vertex_scratch && d_verts && d_faces && d_normals && d_values, "bad argument");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (vertices == 0) return 0;
    const McDims d = make_dims(n0, n1, n2);
    McWorkspace ws;
    carve(d, static_cast<char*>(d_workspace), &ws);
    ws.vertex_cube = static_cast<int64_t*>(d_vertex_scratch);
    ws.vertex_edge = reinterpret_cast<int8_t*>(static_cast<char*>(d_vertex_scratch) + al((size_t)vertices * 8));
    ws.active = reinterpret_cast<McActive*>(static_cast<char*>(d_vertex_scratch) + al((size_t)vertices * 8) + al((size_t)vertices));
    McOut out;
    out.verts = d_verts; out.faces = d_faces; out.normals = d_normals; out.values = d_values;
    for (int a = 0; a < 3; ++a) out.edge_vertex[a] = ws.edge[a];
    const unsigned ablocks = (unsigned)((faces + 255) / 256);   // active cubes <= faces; surplus threads exit
    hipLaunchKernelGGL(mc_compact, dim3((unsigned)((ws.tiles + 3) / 4)), dim3(256), 0, stream, d, ws.codes, ws.tiles,
                       ws.tile_sums, ws.group_sums, ws.active);
    hipLaunchKernelGGL(mc_emit<false>, dim3(ablocks), dim3(256), 0, stream, d_volume, d, iso, ws.active, ws.totals, out,
                       ws.vertex_cube, ws.vertex_edge);
    hipLaunchKernelGGL(mc_emit<true>, dim3(ablocks), dim3(256), 0, stream, d_volume, d, iso, ws.active, ws.totals, out,
                       ws.vertex_cube, ws.vertex_edge);
    hipLaunchKernelGGL(mc_vertex_attributes, dim3((unsigned)((vertices + 255) / 256)), dim3(256), 0, stream, d_volume, d, iso,
                       ws.vertex_cube, ws.vertex_edge, vertices, out);
    NM_HIP_CHECK(hipGetLastError());
    (void)faces;
    return 0;
}

}  // extern "C"
