// Lewiner MC33 marching cubes for gfx950, output-identical to skimage.measure.marching_cubes(volume, level)
// (the call at /root/reference/src/mesh_nerf.py:79: method='lewiner', step_size=1, 'descent', no mask):
// same vertices (bit for bit), same vertex NUMBERING, same triangle order, same normals and values.
//
// The CPU algorithm is a sequential scan (axis0 outermost, axis2 innermost) with a per-layer vertex
// cache; vertex ids are "order of first creation", triangle order is "cube scan order, tiling order".
// Parallel restatement used here (round 3: everything downstream of the volume scan works on the cut cubes only, which are
// kept in scan order from the start -- no per-cube array, nothing to clear, no global queue):
//   1. stream     mc_classify_stream: the HBM-bound pass.  Workgroups of 8 rows x 512 voxels march up axis 0; per voxel the
//                 only work is the sign test, whose 64 results per wavefront are a scalar mask; one wavefront per workgroup
//                 combines the masks of a step into CUT masks (corner bits neither all 0 nor all 1) and writes them to HBM:
//                 8 qwords per UNIT = (plane, row, x brick), units in scan order (64 B per 512 cubes).
//   2. cut        mc_classify_cut, one wavefront per SEGMENT (7 rows of a plane, what a workgroup covers per step): the
//                 masks become the segment's cut cubes IN SCAN ORDER (a lane owns 4 consecutive cubes; the cubes in front
//                 of them are a population count), each gets its MC33 tests -> tiling row, triangle count, number of
//                 vertices it CREATES (an edge vertex is created by the first cube in scan order that contains the edge:
//                 every tiling uses exactly its sign-changing edges, so ownership is a function of the edge position);
//                 one 32-bit entry per cut cube in the unit's slot, (vertices, triangles, cubes) per unit.
//   3. scan       exclusive prefix sums over the units (mc_scan_groups / mc_scan_totals): ~230 k units at 480^3, not
//                 110 M cubes.
//   4. compact    mc_compact: cut cubes -> list (cube id | tiling row, vertex base, triangle base), still in scan order.
//   5. vertices   mc_emit<false>: one thread per cut cube walks its tiling; every first use of an owned edge (or of the
//                 centre vertex, edge id 12) creates vertex base+k: position (fp64 inverse-|v| interpolation rounded to
//                 fp32, as skimage), index stored in an edge->vertex table (int32 volume per axis).
//      faces      mc_emit<true>: walks the tiling again and looks the indices up (no volume access: the tiling row rides in
//                 the list entry); triples are reversed ('descent').
//   6. normals    mc_vertex_attributes, one thread per vertex: re-plays, in scan order, the <= 4 cubes sharing its edge and
//                 accumulates their gradient contributions in fp32 in exactly skimage's order, takes the max of the cubes'
//                 value ranges, normalises in fp64.
// A SLAB of a larger grid (per-rank marching cubes, nm_mc_count_slab / nm_mc_emit_slab) runs the same passes with the
// ownership rule on GLOBAL plane indices and a ghost layer on either side; see McSlab.
// HBM-bound integer/byte work: 4 B/voxel streamed once for classification is the algorithmic traffic
// (442 MB at 480^3); the other passes touch only the ~1 % of cubes that are cut.
#include <math.h>
#include <mutex>

#include "nm_internal.h"
#include "mc_luts.h"

namespace nm {

constexpr double SK_EPS = 2.220446049250313e-16;   // skimage's "FLT_EPSILON" is np.spacing(1.0)
constexpr int MC_ITEMS = 4;                          // CONSECUTIVE cubes per thread (their 4 code bytes = one dword)

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
// Per-edge constants as packed immediates (a table in memory costs a dependent load wherever the edge id is a run-time
// value, and every pass over the cut cubes is a chain of such loads): corners are "bitwise" indices dz*4 + dy*2 + dx
// (skimage's vv[] / vg[] numbering), 3 bits per edge; the axis (0 = x / axis2, 1 = y / axis1, 2 = z / axis0) 2 bits.
constexpr int MC_EDGE_A_[12] = {0, 1, 3, 2, 4, 5, 7, 6, 0, 1, 3, 2};      // end A of edge e (Lewiner numbering)
constexpr int MC_EDGE_B_[12] = {1, 3, 2, 0, 5, 7, 6, 4, 4, 5, 7, 6};      // end B
constexpr int MC_EDGE_LO_[12] = {0, 1, 2, 0, 4, 5, 6, 4, 0, 1, 3, 2};     // the edge's lower corner (the voxel that names the edge)
constexpr int MC_EDGE_AXIS_[12] = {0, 1, 0, 1, 0, 1, 0, 1, 2, 2, 2, 2};
constexpr uint64_t mc_pack(const int (&v)[12], int bits) {
    uint64_t r = 0;
    for (int e = 0; e < 12; ++e) r |= (uint64_t)v[e] << (bits * e);
    return r;
}
constexpr uint64_t MC_PACK_A = mc_pack(MC_EDGE_A_, 3), MC_PACK_B = mc_pack(MC_EDGE_B_, 3), MC_PACK_LO = mc_pack(MC_EDGE_LO_, 3);
constexpr uint32_t MC_PACK_AXIS = (uint32_t)mc_pack(MC_EDGE_AXIS_, 2);
__device__ __forceinline__ int edge_a(int e) { return (int)((MC_PACK_A >> (3 * e)) & 7u); }
__device__ __forceinline__ int edge_b(int e) { return (int)((MC_PACK_B >> (3 * e)) & 7u); }
__device__ __forceinline__ int edge_lo(int e) { return (int)((MC_PACK_LO >> (3 * e)) & 7u); }
__device__ __forceinline__ int edge_axis(int e) { return (int)((MC_PACK_AXIS >> (2 * e)) & 3u); }

// this cube creates the vertex on edge e iff no cube earlier in scan order contains that edge: bit e of the mask
// (bit 12 = the centre vertex of the MC33 tilings that have one)
__device__ __forceinline__ uint32_t owned_edges(int z, int y, int x) {
    const uint32_t z0 = z == 0, y0 = y == 0, x0 = x == 0;
    return ((y0 & z0) << 0) | (z0 << 1) | (z0 << 2) | ((x0 & z0) << 3) | (y0 << 4) | (1u << 5) | (1u << 6) | (x0 << 7) |
           ((x0 & y0) << 8) | (y0 << 9) | (1u << 10) | (x0 << 11) | (1u << 12);
}

// A tiling row (3 bytes per triangle, <= 12 triangles) in registers.  The rows are walked byte by byte in every pass
// over the cut cubes; fetched one byte per step that walk is a chain of 3 nt dependent loads per lane (30 us per
// wavefront in the faces pass, PMC round 3) -- fetched whole it is one round trip (unaligned dword loads; bytes past
// the row belong to the next rows of MC_LUT and are never looked at).
struct TilingRow { uint32_t w[9]; };
static_assert(MC_TEST3_OFF + 36 <= MC_LUT_SIZE, "row loads stay inside MC_LUT (tiling rows end where the test tables begin)");
__device__ __forceinline__ void load_row(int off, int nt, TilingRow& r) {
#pragma unroll
    for (int q = 0; q < 3; ++q) __builtin_memcpy(&r.w[q], MC_LUT + off + 4 * q, 4);
#pragma unroll
    for (int q = 3; q < 9; ++q) r.w[q] = 0u;
    if (nt > 4) {
#pragma unroll
        for (int q = 3; q < 9; ++q) __builtin_memcpy(&r.w[q], MC_LUT + off + 4 * q, 4);
    }
}
__device__ __forceinline__ int row_edge(const TilingRow& r, int i) { return (int)((r.w[i >> 2] >> (8 * (i & 3))) & 0xffu); }   // i: compile-time

// edges the row uses (bit mask)
__device__ __forceinline__ uint32_t row_edges_used(const TilingRow& r, int nt) {
    uint32_t used = 0;
#pragma unroll
    for (int i = 0; i < 12; ++i) used |= i < 3 * nt ? 1u << (row_edge(r, i) & 31) : 0u;
    if (nt > 4) {
#pragma unroll
        for (int i = 12; i < 36; ++i) used |= i < 3 * nt ? 1u << (row_edge(r, i) & 31) : 0u;
    }
    return used;
}

// the vertices a cube creates, in creation order = order of first use in its tiling: 4 bits per edge id, `n` of them
__device__ __forceinline__ uint64_t row_created(const TilingRow& r, int nt, uint32_t owned, int& n) {
    uint32_t seen = 0;
    uint64_t list = 0;
    int cnt = 0;
    auto step = [&](int i) {
        const int e = row_edge(r, i) & 15;
        const uint32_t bit = 1u << e;
        const bool take = i < 3 * nt && !(seen & bit) && (owned & bit);
        list |= take ? (uint64_t)e << (4 * cnt) : 0ull;
        cnt += take ? 1 : 0;
        seen |= bit;
    };
#pragma unroll
    for (int i = 0; i < 12; ++i) step(i);
    if (nt > 4) {
#pragma unroll
        for (int i = 12; i < 36; ++i) step(i);
    }
    n = cnt;
    return list;
}

// code byte per cube: low nibble = triangles (<= 12), high nibble = vertices this cube creates (<= 13).
// The tiling row itself is NOT stored: the ~1 % of cubes that are active re-derive it in the emit passes,
// which keeps the per-cube state at 1 B instead of 4 B (the volume itself is 4 B / voxel).
__device__ __forceinline__ uint32_t pack_code(int nt, int ncreated) { return (uint32_t)nt | ((uint32_t)ncreated << 4); }
__device__ __forceinline__ int code_nt(uint32_t c) { return c & 0xf; }
__device__ __forceinline__ int code_created(uint32_t c) { return (c >> 4) & 0xf; }

// slot entry of a cut cube (32 bits): position in its brick (9) | triangles (4) | created vertices (4) | tiling row = offset into
// MC_LUT (15: the tiling tables end below 2^15).  The tiling row is what the MC33 face / interior tests decide; carrying it
// along means the emit passes do not repeat the tests (and the faces pass does not touch the volume at all).
static_assert(MC_TEST3_OFF < (1 << 15), "tiling rows fit 15 bits");
__device__ __forceinline__ uint32_t pack_entry(uint32_t xoff, int nt, int created, int off) {
    return xoff | ((uint32_t)nt << 9) | ((uint32_t)created << 13) | ((uint32_t)off << 17);
}
__device__ __forceinline__ uint32_t entry_xoff(uint32_t e) { return e & 0x1ffu; }
__device__ __forceinline__ int entry_nt(uint32_t e) { return (e >> 9) & 0xf; }
__device__ __forceinline__ int entry_created(uint32_t e) { return (e >> 13) & 0xf; }
__device__ __forceinline__ int entry_off(uint32_t e) { return (int)(e >> 17); }
// active-cube ids carry the tiling row and triangle count above bit 40 (cube counts are limited to 2^40)
constexpr int64_t MC_ID_MASK = (int64_t(1) << 40) - 1;
__device__ __forceinline__ int64_t pack_id(int64_t id, int nt, int off) { return id | ((int64_t)off << 40) | ((int64_t)nt << 55); }
__device__ __forceinline__ int id_off(int64_t id) { return (int)((id >> 40) & 0x7fff); }
__device__ __forceinline__ int id_nt(int64_t id) { return (int)((id >> 55) & 0xf); }

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
// Two kernels (round 1 did both in one 118-VGPR kernel with a 32 KB LDS queue: 4 workgroups per CU, 0.94 TB/s, 1.42x
// over-fetch; round 2 appended the cut cubes to a global queue with atomics and sorted nothing, which cost a code byte
// per cube -- 110 MB memset + 110 MB scan -- downstream).
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
//     (lane = wave-row x column, 56 lanes) into cut masks with a dozen 64-bit VALU operations -- one pass -- and
//     stores them: 8 qwords per unit (plane, row, brick), 56 lanes x 8 B per step and workgroup.  Nothing else leaves
//     this kernel; the other 15 wavefronts only load and compare.
//     History of this kernel at 480^3 (tests/tools/membw.hip streams the same bytes in the same workgroup shape,
//     barrier per step included, in 76 us; this kernel with the logic removed: 97 us): 8-bit pattern assembled per
//     cube on the VALU (~30 ops / cube) 244 us; mask logic on the scalar unit of every wave (~60 s_and / s_or per
//     wave and step: the CU's one scalar ALU becomes the bottleneck) 154 us; one logic wave per workgroup + owners
//     appending cut cubes to an LDS staging buffer 103 us; masks only (this version) 93 us: see DESIGN.md 3.3.
// (b) mc_classify_cut -- one wavefront per segment (7 rows of a plane x one brick): turns the 56 masks into the
//     segment's cut cubes in scan order, runs the MC33 face / interior tests (fp64) for each, and writes one packed
//     entry per cut cube (x offset, triangle count, vertices created, tiling row) into the unit's slot.
constexpr int MC_ZRUN_MIN = 8;                     // planes per march: chosen per launch (mc_pick_zrun)
constexpr int MC_GX = 128, MC_GY = 8;                               // thread grid: x-groups per row, rows (7 cube rows + halo)
constexpr int MC_STREAM_THREADS = MC_GX * MC_GY;
constexpr int MC_UNIT = MC_GX * MC_ITEMS;                            // cubes of one row that one workgroup covers (512)
constexpr int MC_AHEAD = 4;   // even (LDS double buffers are indexed by the ring slot's parity).  6 needs 67 VGPRs: 3 spills, one
                              // of them reloaded -- behind an s_waitcnt vmcnt(0), i.e. behind all its prefetches -- by the logic wave every step
constexpr bool MC_MARCH0 = true;   // march along axis 0, thread rows = adjacent rows of axis 1 (false: the other way round; same speed)
constexpr int MC_LOGIC_WAVE = MC_STREAM_THREADS / 64 - 1;           // second half of the halo row: owns no cubes
static_assert(MC_GX == 128 && MC_ITEMS == 4 && 8 * (MC_GY - 1) <= 64 && MC_AHEAD % 2 == 0, "two wavefronts per row, four voxels per lane");


// Sign masks of one row as a wavefront sees it: mask j, bit L = voxel 4 L + j of the wave's 256-voxel span; "column 4"
// (voxel 4 L + 4) is column 0 shifted down one lane with the neighbour wave's / halo voxel's bit on top.
template <bool VEC, bool XHALO>   // VEC: n2 % 4 == 0 (every x-group is one 16-byte load); XHALO: more than one brick in x
__global__ __launch_bounds__(MC_STREAM_THREADS, 8) void mc_classify_stream(const float* __restrict__ vol, McDims d, float thr,
                                                                        uint64_t* __restrict__ cut_masks, const int zrun) {
    __shared__ uint64_t s_m[2][MC_GY][2][4];             // sign masks   [step parity][row][x half][column]
    __shared__ uint64_t s_prev[2 * (MC_GY - 1)][2][5];   // logic wave: [wave-row][AND, OR][column] of the previous step
    __shared__ uint32_t s_h[2][MC_GY];                   // the voxel behind the brick
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
    // logic wave: lane = 4 k + column, k = 2 cube row + x half (56 lanes)
    const int kk = lane >> 2, kcol = lane & 3, krow = min(kk >> 1, MC_GY - 2), khalf = kk & 1;
    const bool k_exists = r0 + krow < cr;
    // cubes of column kcol that exist in this wave-row: bit L = cube 4 L + kcol of the half-row (lanes past the end of a row
    // hold copies of its last voxels)
    const int k_valid = max(0, min(MC_UNIT / 2, d.c2 - xb - khalf * (MC_UNIT / 2)));
    const int k_lanes = max(0, min(64, (k_valid - kcol + 3) >> 2));
    const uint64_t k_colmask = k_lanes >= 64 ? ~0ull : ((1ull << k_lanes) - 1ull);
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
        __syncthreads();
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
                if (l > 0 && k_exists && m0 + l - 1 < cm) cut = any1 & ~all1 & k_colmask;      // cubes of step l - 1
                s_prev[kk][0][kcol] = and_c;
                s_prev[kk][1][kcol] = or_c;
                if (kcol == 3) { s_prev[kk][0][4] = and_n; s_prev[kk][1][4] = or_n; }
                // The cut mask of the 64 cubes 4 L + kcol of half-row (krow, khalf), step l - 1, goes straight to HBM: 8
                // qwords per unit = (plane, row, x brick), units in scan order.  Every existing unit is written, cut or not
                // (nothing is cleared beforehand); the next pass turns the masks into ordered entries.
                if (l > 0 && k_exists && m0 + l - 1 < cm) {
                    const int cz = MC_MARCH0 ? m0 + l - 1 : r0 + krow, cy = MC_MARCH0 ? r0 + krow : m0 + l - 1;
                    cut_masks[(((int64_t)cz * d.c1 + cy) * gridDim.x + blockIdx.x) * 8 + khalf * 4 + kcol] = cut;
                }
            }
        }
      }
    }
}

// Planes per march: 24 (halo planes cost 25/24; 1380 workgroups at 480^3), fewer for volumes that would not give every
// CU its two workgroups.  (Choosing zrun to make the launch a whole number of "rounds" -- 22 at 480^3 -- measures
// slower, 124.5 vs 121 us: workgroups do not run in lockstep rounds, the extra halo planes are what counts.)
// (measured at 480^3, round 3: 24 planes -> 96 us, 35 -> 92, 48 -> 106, 60 -> 124, 69 = one resident round -> 92, 120 -> 127:
// the pass is not limited by the tail of its last round)
static int mc_pick_zrun(int64_t bricks_xy, int c_march, int num_cus) {
    const int64_t slots = 2 * (int64_t)(num_cus > 0 ? num_cus : 256);
    int z = 24;
    while (z > MC_ZRUN_MIN && bricks_xy * ((c_march + z - 1) / z) < slots) --z;
    return z;
}

// A SEGMENT is what one workgroup of mc_classify_stream covers in one step: MC_GY - 1 = 7 consecutive rows of one plane,
// one x brick wide = 7 units.  The passes over the cut cubes run one wavefront per segment (most segments are empty and
// exit on their counts), lanes = cut cubes in (row, position) order.
struct McSegment {
    int z, r0, xb;             // plane, first row, x brick
    int64_t unit0, ustride;    // unit of row r0; + ustride per row
    uint32_t start[MC_GY];     // start[t] = cut cubes in rows r0 .. r0 + t - 1; start[7] = all of them
};

__device__ __forceinline__ void mc_segment_place(const McDims& d, int64_t seg, int bx, int by, McSegment& s) {
    s.xb = (int)(seg % bx);
    const int64_t zy = seg / bx;
    s.r0 = (int)(zy % by) * (MC_GY - 1);
    s.z = (int)(zy / by);
    s.unit0 = ((int64_t)s.z * d.c1 + s.r0) * bx + s.xb;
    s.ustride = bx;
}

// row t of entry e (start[] is non-decreasing) and the entry's position in its row
__device__ __forceinline__ int mc_segment_row(const McSegment& s, uint32_t e, uint32_t& k) {
    int t = 0;
#pragma unroll
    for (int q = 1; q < MC_GY - 1; ++q) t += e >= s.start[q] ? 1 : 0;
    uint32_t st = s.start[0];
#pragma unroll
    for (int q = 1; q < MC_GY - 1; ++q) st = t == q ? s.start[q] : st;
    k = e - st;
    return t;
}

// (Measured and dropped, round 3: the case / test tables of the MC33 decision staged in LDS by every working wavefront --
// 129 instead of 98 VGPRs, 58 us instead of 49.)
// (b) mc_classify_cut: the cut masks of a segment -> its cut cubes in scan order -> their MC33 tests.  Per unit, the slot
// receives one entry per cut cube (pack_entry: position in the brick | triangles | created vertices | tiling row) and unit_sums the
// unit's (created vertices, triangles, cut cubes).  Every unit is written (zeros for the untouched ones).
//   Lane L of a half-row holds cubes 4 L .. 4 L + 3, mask j bit L = cube 4 L + j: the number of cut cubes in front of a
//   lane's four is the population count of the four masks below bit L, so a wave lists a row in cube order with 8 mbcnt.
constexpr int MC_SEG_MAX = (MC_GY - 1) * MC_UNIT;      // 3584 cubes per segment
__global__ __launch_bounds__(256) void mc_classify_cut(const float* __restrict__ vol, McDims d, double iso, int zg0,
                                                       const uint64_t* __restrict__ cut_masks, uint32_t* __restrict__ slots,
                                                       const int cap, int bx, int by, int64_t segments,
                                                       uint4* __restrict__ unit_sums, uint32_t* __restrict__ cube_entries) {
    __shared__ uint16_t s_x[4][MC_SEG_MAX];             // position in the brick of every cut cube of the segment, in order
    __shared__ uint32_t s_acc[4][MC_GY - 1][2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t seg = (int64_t)blockIdx.x * 4 + wave;
    if (seg >= segments) return;
    McSegment sg;
    mc_segment_place(d, seg, bx, by, sg);
    // lanes 8 t + 4 h + j: mask j of half h of row t
    const int mt = lane >> 3;
    const bool row_exists = mt < MC_GY - 1 && sg.r0 + mt < d.c1;
    uint64_t mask = 0;
    if (row_exists) mask = cut_masks[(sg.unit0 + mt * sg.ustride) * 8 + (lane & 7)];
    uint32_t cnt = (uint32_t)__popcll(mask);
    cnt += __shfl_xor(cnt, 1); cnt += __shfl_xor(cnt, 2); cnt += __shfl_xor(cnt, 4);        // the row's cut cubes, in its 8 lanes
    const uint32_t row_cnt = (uint32_t)__shfl((int)cnt, lane < MC_GY - 1 ? 8 * lane : 0);   // lane t: cut cubes of row t
    uint32_t run = 0;
#pragma unroll
    for (int t = 0; t < MC_GY - 1; ++t) { sg.start[t] = run; run += (uint32_t)__builtin_amdgcn_readlane((int)cnt, 8 * t); }
    sg.start[MC_GY - 1] = run;
    const uint32_t total = run;
    if (lane < MC_GY - 1 && sg.r0 + lane < d.c1 && total == 0) unit_sums[sg.unit0 + lane * sg.ustride] = make_uint4(0u, 0u, 0u, 0u);
    if (total == 0) return;
    if (lane < 2 * (MC_GY - 1)) s_acc[wave][lane >> 1][lane & 1] = 0;
    // ---- the segment's cut cubes in order
    const unsigned long long below = (1ull << lane) - 1ull;
    const int mlo = (int)(uint32_t)mask, mhi = (int)(uint32_t)(mask >> 32);
#pragma unroll
    for (int t = 0; t < MC_GY - 1; ++t) {
        if (sg.start[t + 1] == sg.start[t]) continue;
        uint32_t pos = sg.start[t];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            uint64_t m[4];
            uint32_t in_front = 0, own = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int src = 8 * t + 4 * h + j;
                m[j] = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane(mhi, src) << 32) | (uint32_t)__builtin_amdgcn_readlane(mlo, src);
                in_front += (uint32_t)__popcll(m[j] & below);
                own += (uint32_t)__popcll(m[j]);
            }
            uint32_t p = pos + in_front;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if ((m[j] >> lane) & 1ull) s_x[wave][p++] = (uint16_t)(h * (MC_UNIT / 2) + 4 * lane + j);
            pos += own;
        }
    }
    __builtin_amdgcn_wave_barrier();
    // ---- the MC33 tests, one lane per cut cube
    for (uint32_t e = lane; e < total; e += 64) {
        uint32_t k;
        const int t = mc_segment_row(sg, e, k);
        const uint32_t xoff = s_x[wave][e];
        const int z = sg.z, y = sg.r0 + t, x = sg.xb * MC_UNIT + (int)xoff;
        Cube c;
        load_cube(vol, d, z, y, x, iso, c);
        int off, nt;
        select_tiling(c, index_of(c), off, nt);
        TilingRow row;
        load_row(off, nt, row);
        const int created = __popc(row_edges_used(row, nt) & owned_edges(z + zg0, y, x));
        const uint32_t entry = pack_entry(xoff, nt, created, off);
        slots[(sg.unit0 + t * sg.ustride) * cap + k] = entry;       // in scan order, for the numbering passes
        cube_entries[((int64_t)z * d.c1 + y) * d.c2 + x] = entry;   // by position, for the attribute pass
        atomicAdd(&s_acc[wave][t][0], (uint32_t)created);
        atomicAdd(&s_acc[wave][t][1], (uint32_t)nt);
    }
    __builtin_amdgcn_wave_barrier();
    if (lane < MC_GY - 1 && sg.r0 + lane < d.c1)
        unit_sums[sg.unit0 + lane * sg.ustride] =
            make_uint4(s_acc[wave][lane][0], s_acc[wave][lane][1], row_cnt, 0u);
}

// ---- pass 2: exclusive scan of the per-unit sums (units are in scan order): 1024-tile groups in parallel, then the group totals ------
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

// `host` (pinned, device-mapped): (vertices, triangles, cut cubes, vertices and triangles in front of unit `ghost_unit`) for
// the caller, who only has to wait for the stream -- no copy to stage
__global__ __launch_bounds__(1024) void mc_scan_totals(uint4* __restrict__ group_sums, int64_t groups,
                                                       uint32_t* __restrict__ totals, const uint4* __restrict__ unit_prefix,
                                                       int64_t ghost_unit, uint32_t* __restrict__ host) {
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
    if (threadIdx.x == 0) {
        totals[0] = carry[0]; totals[1] = carry[1]; totals[2] = carry[2];
        uint32_t gv = 0, gf = 0;
        if (ghost_unit >= 0) {     // exclusive prefix of that unit = what the layers in front of it account for
            const uint4 u = unit_prefix[ghost_unit], g = group_sums[ghost_unit >> 10];
            gv = u.x + g.x; gf = u.y + g.y;
        }
        host[0] = carry[0]; host[1] = carry[1]; host[2] = carry[2]; host[3] = gv; host[4] = gf;
    }
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

// ---- pass 3a: the cut cubes (scan order) with their vertex / triangle bases ------------------------------------------
// one wavefront per segment; a unit's entries are consecutive lanes, so the running sums of (created, triangles) over the
// segment's entries minus their value at the unit's first entry are the in-unit prefixes; the unit's own base comes from
// the scan over the units (which are in scan order: (plane, row, x brick)).
__global__ __launch_bounds__(256) void mc_compact(McDims d, const uint32_t* __restrict__ slots, const int cap, int bx, int by,
                                                  int64_t segments, const uint4* __restrict__ unit_prefix,
                                                  const uint4* __restrict__ group_prefix, McActive* __restrict__ list,
                                                  const uint32_t* __restrict__ totals, uint32_t list_capacity) {
    __shared__ uint32_t s_first[4][MC_GY - 1][2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t seg = (int64_t)blockIdx.x * 4 + wave;
    if (seg >= segments) return;
    if (totals[1] > list_capacity) return;     // launched ahead of the host's look at the totals: the list may not fit (see nm_mc_emit_slab)
    // counts of the 7 units: .w of the scanned entries (mc_scan_groups keeps each unit's own count there)
    McSegment sg;
    mc_segment_place(d, seg, bx, by, sg);
    {
        uint32_t c = 0;
        if (lane < MC_GY - 1 && sg.r0 + lane < d.c1) c = unit_prefix[sg.unit0 + lane * sg.ustride].w;
        uint32_t run = 0;
#pragma unroll
        for (int t = 0; t < MC_GY - 1; ++t) { sg.start[t] = run; run += (uint32_t)__builtin_amdgcn_readlane((int)c, t); }
        sg.start[MC_GY - 1] = run;
        if (run == 0) return;
    }
    const uint32_t total = sg.start[MC_GY - 1];
    uint32_t carry_v = 0, carry_t = 0;
    for (uint32_t base = 0; base < total; base += 64) {
        const uint32_t e = base + lane;
        const bool live = e < total;
        int t = 0;
        uint32_t k = 0, ent = 0;
        if (live) {
            t = mc_segment_row(sg, e, k);
            ent = slots[(sg.unit0 + t * sg.ustride) * cap + k];
        }
        uint32_t iv = live ? entry_created(ent) : 0u, it = live ? entry_nt(ent) : 0u;
        const uint32_t ov = iv, ot = it;
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t pv = __shfl_up(iv, o), pt = __shfl_up(it, o);
            if (lane >= o) { iv += pv; it += pt; }
        }
        const uint32_t ev = carry_v + iv - ov, et = carry_t + it - ot;          // exclusive over the segment's entries
        if (live && k == 0) { s_first[wave][t][0] = ev; s_first[wave][t][1] = et; }
        __builtin_amdgcn_wave_barrier();
        if (live) {
            const int64_t unit = sg.unit0 + t * sg.ustride;
            const uint4 up = unit_prefix[unit], gp = group_prefix[unit >> 10];
            McActive a;
            a.id = pack_id(((int64_t)sg.z * d.c1 + (sg.r0 + t)) * d.c2 + (sg.xb * MC_UNIT + (int)entry_xoff(ent)), entry_nt(ent), entry_off(ent));
            a.vbase = gp.x + up.x + (ev - s_first[wave][t][0]);
            a.tbase = gp.y + up.y + (et - s_first[wave][t][1]);
            list[gp.z + up.z + k] = a;
        }
        carry_v += (uint32_t)__builtin_amdgcn_readlane((int)iv, 63);
        carry_t += (uint32_t)__builtin_amdgcn_readlane((int)it, 63);
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- pass 3b: vertices and faces, one thread per ACTIVE cube (all lanes busy) ---------------------------------
// A SLAB of a larger volume (per-rank marching cubes, see nm_mc_count_slab): local plane 0 is global plane zg0; the
// first `ghost` cube layers belong to the slab below -- their cubes are classified and numbered (the faces of the layer above
// reference the vertices they create) but emit nothing: vertex / face rows start behind the ghost_v / ghost_f they
// account for, and vertex ids are written as local id + index_base.
struct McSlab {
    int zg0, ghost;
    uint32_t ghost_v, ghost_f;
    int64_t index_base;
};

// two triangles of a cube's tiling (2 TP, 2 TP + 1): six table reads in flight together, two 12-byte rows out
template <int TP>
__device__ __forceinline__ void emit_pair(const TilingRow& row, int nt, int centre, int32_t ib, int z, int y, int x,
                                          const McDims& d, const McOut& out, int32_t* f) {
    const bool two = 2 * TP + 1 < nt;
    int idx[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int e = (two || j < 3) ? row_edge(row, 6 * TP + j) : row_edge(row, 6 * TP + (j < 3 ? j : j - 3));
        const int ee = e == 12 ? 0 : (e & 15), lo = edge_lo(ee), axis = edge_axis(ee);
        const int64_t vox = ((int64_t)(z + (lo >> 2)) * d.n1 + (y + ((lo >> 1) & 1))) * d.n2 + (x + (lo & 1));
        const int32_t* table = axis == 0 ? out.edge_vertex[0] : (axis == 1 ? out.edge_vertex[1] : out.edge_vertex[2]);
        const int v = table[vox];
        idx[j] = e == 12 ? centre : v;
    }
    int32_t* g = f + 6 * TP;                      // gradient_direction='descent' reverses each triple
    g[0] = idx[2] + ib; g[1] = idx[1] + ib; g[2] = idx[0] + ib;
    if (two) { g[3] = idx[5] + ib; g[4] = idx[4] + ib; g[5] = idx[3] + ib; }
}

template <bool FACES>
__global__ __launch_bounds__(256) void mc_emit(const float* __restrict__ vol, McDims d, double iso, McSlab sl,
                                               const McActive* __restrict__ list, const uint32_t* __restrict__ totals,
                                               McOut out, int64_t* __restrict__ vertex_cube,
                                               int8_t* __restrict__ vertex_edge) {
    const int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= (int64_t)totals[2]) return;     // number of active cubes, left on the device by the scan
    const McActive ent = list[a];
    int z, y, x;
    cube_coords(d, ent.id & MC_ID_MASK, z, y, x);
    const int off = id_off(ent.id), nt = id_nt(ent.id);        // the tiling row mc_classify_cut chose
    TilingRow row;
    load_row(off, nt, row);
    int ncreated;
    const uint64_t created = row_created(row, nt, owned_edges(z + sl.zg0, y, x), ncreated);
    if constexpr (!FACES) {
        Cube c;
        load_cube(vol, d, z, y, x, iso, c);
        for (int k = 0; k < ncreated; ++k) {
            const int e = (int)((created >> (4 * k)) & 15u);
            const uint32_t next = ent.vbase + (uint32_t)k;
            double px, py, pz;
            if (e == 12) {
                double fx = 0, fy = 0, fz = 0, ff = 0;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const double w = 1.0 / (SK_EPS + fabs(c.v[q]));
                    const int cx = (q == 1 || q == 2 || q == 5 || q == 6), cy = (q == 2 || q == 3 || q == 6 || q == 7), cz = q >> 2;
                    fx += (double)cx * w; fy += (double)cy * w; fz += (double)cz * w; ff += w;
                }
                px = x + fx / ff; py = y + fy / ff; pz = (z + sl.zg0) + fz / ff;
            } else {
                const int ea = edge_a(e), eb = edge_b(e), lo = edge_lo(e);    // dz*4 + dy*2 + dx
                const int ka = e < 8 ? ((e & 3)) + (e & 4) : e - 8;           // Lewiner corner of end A
                const int kb = e < 8 ? (((e & 3) + 1) & 3) + (e & 4) : e - 4; // Lewiner corner of end B
                const double w1 = 1.0 / (SK_EPS + fabs(pick(c, ka))), w2 = 1.0 / (SK_EPS + fabs(pick(c, kb)));
                double fx = 0, fy = 0, fz = 0, ff = 0;
                fx += (double)(ea & 1) * w1; fy += (double)((ea >> 1) & 1) * w1; fz += (double)(ea >> 2) * w1; ff += w1;
                fx += (double)(eb & 1) * w2; fy += (double)((eb >> 1) & 1) * w2; fz += (double)(eb >> 2) * w2; ff += w2;
                px = x + fx / ff; py = y + fy / ff; pz = (z + sl.zg0) + fz / ff;
                const int64_t vox = ((int64_t)(z + (lo >> 2)) * d.n1 + (y + ((lo >> 1) & 1))) * d.n2 + (x + (lo & 1));
                const int axis = edge_axis(e);
                int32_t* table = axis == 0 ? out.edge_vertex[0] : (axis == 1 ? out.edge_vertex[1] : out.edge_vertex[2]);
                table[vox] = (int32_t)next;
            }
            if (z >= sl.ghost) {
                const int64_t r = (int64_t)next - sl.ghost_v;
                // wrapper: vertices flipped to (axis0, axis1, axis2) = (z, y, x)
                out.verts[3 * r] = (float)pz;
                out.verts[3 * r + 1] = (float)py;
                out.verts[3 * r + 2] = (float)px;
                vertex_cube[r] = ent.id;
                vertex_edge[r] = (int8_t)e;
            }
        }
    } else {
        if (z < sl.ghost) return;
        // the centre vertex (if any) is created by this cube: its id = vbase + its place in the creation order
        int centre = -1;
#pragma unroll
        for (int k = 0; k < 13; ++k) centre = (k < ncreated && ((created >> (4 * k)) & 15u) == 12u) ? (int)ent.vbase + k : centre;
        const int32_t ib = (int32_t)sl.index_base;
        int32_t* f = out.faces + 3 * ((int64_t)ent.tbase - sl.ghost_f);
        // two triangles per step: their six table reads are in flight together (a lone last triangle reads its own
        // entries twice); the steps are unrolled so that the row bytes are register selects
        emit_pair<0>(row, nt, centre, ib, z, y, x, d, out, f);
        if (nt > 2) emit_pair<1>(row, nt, centre, ib, z, y, x, d, out, f);
        if (nt > 4) {
            emit_pair<2>(row, nt, centre, ib, z, y, x, d, out, f);
            if (nt > 6) emit_pair<3>(row, nt, centre, ib, z, y, x, d, out, f);
            if (nt > 8) emit_pair<4>(row, nt, centre, ib, z, y, x, d, out, f);
            if (nt > 10) emit_pair<5>(row, nt, centre, ib, z, y, x, d, out, f);
        }
    }
}

// skimage's per-corner gradients (forward differences along the cube edges)
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

// What mc_classify_cut decided for cube (z, y, x): its entry in the per-cube table (written for cut cubes only and never
// cleared: whether the cube IS cut comes from its bit in the cut masks -- mask 4 h + j, bit L <-> cube 256 h + 4 L + j of the
// brick).  Both reads are issued together: one round trip instead of repeating the MC33 tests.  0 = not cut.
struct McLookup { const uint64_t* masks; const uint32_t* entries; int bx; };
__device__ __forceinline__ uint32_t cube_entry(const McLookup& lk, const McDims& dc, int z, int y, int x) {
    const int64_t row = (int64_t)z * dc.c1 + y;
    const int xo = x % MC_UNIT;
    const uint64_t m = lk.masks[(row * lk.bx + x / MC_UNIT) * 8 + ((xo >> 8) * 4 + (xo & 3))];
    const uint32_t ent = lk.entries[row * dc.c2 + x];
    return (m >> ((xo & 255) >> 2)) & 1ull ? ent : 0u;
}

// ---- pass 5: normals + values (one thread per vertex, replaying the <= 4 cubes around its edge in scan order) ----------
// The cubes sharing an edge, as offsets (dz, dy, dx) from the edge's LOWER CORNER voxel, in scan order, with the edge's
// local id inside each:   x edge {-1,-1,0, 6} {-1,0,0, 4} {0,-1,0, 2} {0,0,0, 0}
//                         y edge {-1,0,-1, 5} {-1,0,0, 7} {0,0,-1, 1} {0,0,0, 3}
//                         z edge {0,-1,-1,10} {0,-1,0,11} {0,0,-1, 9} {0,0,0, 8}
// (selected per axis from immediates: k is a compile-time index once the loop over the cubes is unrolled).
struct EdgeEnds { signed char a[3][4], b[3][4]; bool a_is_lo[3][4]; };     // [axis][k]: end corners (bitwise index) of the shared edge inside cube k
constexpr EdgeEnds edge_ends_of_shared_cubes() {
    constexpr signed char E[3][4] = {{6, 4, 2, 0}, {5, 7, 1, 3}, {10, 11, 9, 8}};     // the local edge ids of shared_cube below
    EdgeEnds t{};
    for (int ax = 0; ax < 3; ++ax)
        for (int k = 0; k < 4; ++k) {
            t.a[ax][k] = (signed char)MC_EDGE_A_[E[ax][k]];
            t.b[ax][k] = (signed char)MC_EDGE_B_[E[ax][k]];
            t.a_is_lo[ax][k] = MC_EDGE_A_[E[ax][k]] == MC_EDGE_LO_[E[ax][k]];
        }
    return t;
}

__device__ __forceinline__ void shared_cube(int axis, int k, int& dz, int& dy, int& dx, int& e) {
    constexpr signed char S[3][4][4] = {{{-1, -1, 0, 6}, {-1, 0, 0, 4}, {0, -1, 0, 2}, {0, 0, 0, 0}},
                                        {{-1, 0, -1, 5}, {-1, 0, 0, 7}, {0, 0, -1, 1}, {0, 0, 0, 3}},
                                        {{0, -1, -1, 10}, {0, -1, 0, 11}, {0, 0, -1, 9}, {0, 0, 0, 8}}};
    dz = axis == 0 ? S[0][k][0] : (axis == 1 ? S[1][k][0] : S[2][k][0]);
    dy = axis == 0 ? S[0][k][1] : (axis == 1 ? S[1][k][1] : S[2][k][1]);
    dx = axis == 0 ? S[0][k][2] : (axis == 1 ? S[1][k][2] : S[2][k][2]);
    e = axis == 0 ? S[0][k][3] : (axis == 1 ? S[1][k][3] : S[2][k][3]);
}

__global__ __launch_bounds__(256) void mc_vertex_attributes(const float* __restrict__ vol, McDims d, McDims dc, double iso,
                                                            McLookup lk, const int64_t* __restrict__ vertex_cube,
                                                            const int8_t* __restrict__ vertex_edge, int64_t nverts,
                                                            McOut out) {
    const int64_t vid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (vid >= nverts) return;
    const int64_t home = vertex_cube[vid];
    const int e_home = vertex_edge[vid];
    int hz, hy, hx;
    cube_coords(dc, home & MC_ID_MASK, hz, hy, hx);      // ids count the classified layers; the neighbours may lie in the ghost layer above (d)
    float nx = 0.0f, ny = 0.0f, nz = 0.0f, value = 0.0f;
    int axis = 0, lz = hz, ly = hy, lx = hx;
    if (e_home != 12) {
        const int lo = edge_lo(e_home);
        axis = edge_axis(e_home);
        lz = hz + (lo >> 2); ly = hy + ((lo >> 1) & 1); lx = hx + (lo & 1);
    }
    // ---- the cubes around the edge (the centre vertex: its own cube, as k = 3) and what the classification pass chose for
    // them: all four lookups are in flight together
    int cz[4], cy[4], cx[4], ce[4], coff[4], cnt[4];
    bool live[4], tests[4];
    uint32_t entry[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int dz, dy, dx, e;
        shared_cube(axis, k, dz, dy, dx, e);
        if (e_home == 12) { dz = dy = dx = 0; e = 12; }
        cz[k] = lz + dz; cy[k] = ly + dy; cx[k] = lx + dx; ce[k] = e;
        live[k] = (e_home != 12 || k == 3) && cz[k] >= 0 && cy[k] >= 0 && cx[k] >= 0 && cz[k] < d.c0 && cy[k] < d.c1 && cx[k] < d.c2;
        tests[k] = live[k] && cz[k] >= dc.c0;           // ghost layer above: classified by the next slab, not here
        entry[k] = 0u;
        if (live[k] && !tests[k]) entry[k] = cube_entry(lk, dc, cz[k], cy[k], cx[k]);
    }
    TilingRow row[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        coff[k] = entry_off(entry[k]); cnt[k] = entry_nt(entry[k]);
        load_row(coff[k], cnt[k], row[k]);
    }
    // `strength` of the edge's two ends, 1 / (eps + |v|) rounded to a C float as in skimage: the same two voxels in every
    // cube that shares the edge (a cube may run the edge the other way round), so the two fp64 divisions are done once per
    // vertex instead of once per cube
    float s_lo = 0.0f, s_hi = 0.0f;
    if (e_home != 12) {
        const int64_t s1v = d.n2, s0v = (int64_t)d.n1 * d.n2;
        const float* pl = vol + (int64_t)lz * s0v + (int64_t)ly * s1v + lx;
        const double v_lo = (double)pl[0] - iso, v_hi = (double)pl[axis == 0 ? 1 : (axis == 1 ? s1v : s0v)] - iso;
        s_lo = (float)(1.0 / (SK_EPS + fabs(v_lo)));
        s_hi = (float)(1.0 / (SK_EPS + fabs(v_hi)));
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (!live[k]) continue;
        const int z = cz[k], y = cy[k], x = cx[k], e = ce[k];
        Cube c;
        load_cube(vol, d, z, y, x, iso, c);
        int off = coff[k], nt = cnt[k];
        if (tests[k]) {
            select_tiling(c, index_of(c), off, nt);
            load_row(off, nt, row[k]);
        }
        if (nt == 0) continue;
        // how often the cube's tiling references the edge
        int uses = 0;
#pragma unroll
        for (int i = 0; i < 12; ++i) uses += (i < 3 * nt && row_edge(row[k], i) == e) ? 1 : 0;
        if (nt > 4) {
#pragma unroll
            for (int i = 12; i < 36; ++i) uses += (i < 3 * nt && row_edge(row[k], i) == e) ? 1 : 0;
        }
        if (uses == 0) continue;
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
            // the cube's local edge id is one of three compile-time values once k is unrolled (one per axis, shared_cube):
            // its end corners are a 3-way select on the axis instead of an 8-way select on the corner (96 -> 24 v_cndmask)
            constexpr EdgeEnds ET = edge_ends_of_shared_cubes();
            const int qa0 = ET.a[0][k], qa1 = ET.a[1][k], qa2 = ET.a[2][k], qb0 = ET.b[0][k], qb1 = ET.b[1][k], qb2 = ET.b[2][k];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                ga[a] = axis == 0 ? g[3 * qa0 + a] : (axis == 1 ? g[3 * qa1 + a] : g[3 * qa2 + a]);
                gb[a] = axis == 0 ? g[3 * qb0 + a] : (axis == 1 ? g[3 * qb1 + a] : g[3 * qb2 + a]);
            }
            const bool a_is_lo = axis == 0 ? ET.a_is_lo[0][k] : (axis == 1 ? ET.a_is_lo[1][k] : ET.a_is_lo[2][k]);   // end A = the edge's lower voxel?
            s1 = a_is_lo ? s_lo : s_hi;
            s2 = a_is_lo ? s_hi : s_lo;
        }
        for (int u = 0; u < uses; ++u) {
            if (e == 12) { nx += gx; ny += gy; nz += gz; }
            else {
                nx += (float)(ga[0] * (double)s1); ny += (float)(ga[1] * (double)s1); nz += (float)(ga[2] * (double)s1);
                nx += (float)(gb[0] * (double)s2); ny += (float)(gb[1] * (double)s2); nz += (float)(gb[2] * (double)s2);
            }
        }
        if (vmax > (double)value) value = (float)vmax;
    }
    const double len = sqrt((double)nx * nx + (double)ny * ny + (double)nz * nz);
    if (len > 0.0) { nx = (float)(nx / len); ny = (float)(ny / len); nz = (float)(nz / len); }
    out.normals[3 * vid] = nz; out.normals[3 * vid + 1] = ny; out.normals[3 * vid + 2] = nx;   // flipped columns
    out.values[vid] = value;
}

static inline size_t al(size_t x) { return (x + 255) & ~size_t(255); }

struct McWorkspace {
    uint64_t* masks; uint32_t* slots; uint32_t* entries; McActive* list; uint32_t list_capacity; uint4* unit_sums; uint4* group_sums; uint32_t* totals; int32_t* edge[3]; int64_t* vertex_cube; int8_t* vertex_edge;
    McActive* active;
    int64_t units;     // (plane, row, x brick) triples, in scan order
    int cap, bx, by;   // entries per unit; x bricks per row; row groups per plane
};

static McDims make_dims(int n0, int n1, int n2) {
    McDims d{n0, n1, n2, n0 - 1, n1 - 1, n2 - 1, (int64_t)(n0 - 1) * (n1 - 1) * (n2 - 1)};
    return d;
}

static size_t carve(const McDims& d, char* base, McWorkspace* ws) {
    const int bx = (d.c2 + MC_UNIT - 1) / MC_UNIT, by = (d.c1 + MC_GY - 2) / (MC_GY - 1);
    const int cap = bx > 1 ? MC_UNIT : ((d.c2 + 3) & ~3);
    const int64_t units = (int64_t)d.c0 * d.c1 * bx;
    const size_t vox = (size_t)d.n0 * d.n1 * d.n2;
    size_t off = 0;
    auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += al(bytes); return p; };
    char* p;
    p = take((size_t)units * 64); if (ws) ws->masks = (uint64_t*)p;
    p = take((size_t)units * cap * 4); if (ws) ws->slots = (uint32_t*)p;
    p = take((size_t)d.cubes * 4); if (ws) ws->entries = (uint32_t*)p;
    // the cut cubes in scan order, built by the count call while the host is still waiting for the totals -- if the surface
    // has at most cubes / 16 triangles (a closed surface cuts ~1 % of a grid's cubes, two triangles each); beyond that the
    // emit call builds the list in its own scratch, as it did before round 3
    const size_t list_capacity = (size_t)(d.cubes / 16 > 65536 ? d.cubes / 16 : 65536);
    p = take(list_capacity * sizeof(McActive)); if (ws) { ws->list = (McActive*)p; ws->list_capacity = (uint32_t)(list_capacity < 0xffffffffu ? list_capacity : 0xffffffffu); }
    p = take((size_t)units * 16); if (ws) ws->unit_sums = (uint4*)p;
    p = take((size_t)((units + 1023) / 1024) * 16); if (ws) ws->group_sums = (uint4*)p;
    p = take(256); if (ws) ws->totals = (uint32_t*)p;
    for (int a = 0; a < 3; ++a) { p = take(vox * 4); if (ws) ws->edge[a] = (int32_t*)p; }
    if (ws) { ws->units = units; ws->cap = cap; ws->bx = bx; ws->by = by; }
    return off;
}

}  // namespace nm

using namespace nm;

// Where the count pass leaves its totals for the host: a ring of 64-byte slots in pinned, device-mapped memory per device,
// each with the event the host waits on (concurrent calls on different streams take different slots; 64 of them in flight
// is far beyond any caller).
struct McHostRing { uint32_t* host = nullptr; uint32_t* dev = nullptr; hipEvent_t done[64] = {}; unsigned next = 0; };
static bool mc_host_slot(uint32_t** host, uint32_t** dev, hipEvent_t* done) {
    static McHostRing per_device[16];
    static std::mutex lock;
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 16) return false;
    std::lock_guard<std::mutex> hold(lock);
    McHostRing& r = per_device[d];
    if (!r.host) {
        void* h = nullptr;
        void* dv = nullptr;
        if (hipHostMalloc(&h, 64 * 64, hipHostMallocMapped) != hipSuccess) { (void)hipGetLastError(); return false; }
        if (hipHostGetDevicePointer(&dv, h, 0) != hipSuccess) { (void)hipGetLastError(); (void)hipHostFree(h); return false; }
        r.host = static_cast<uint32_t*>(h); r.dev = static_cast<uint32_t*>(dv);
    }
    const unsigned k = r.next++ % 64u;
    if (!r.done[k] && hipEventCreateWithFlags(&r.done[k], hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return false; }
    *host = r.host + 16 * k; *dev = r.dev + 16 * k; *done = r.done[k];
    return true;
}

// Side stream for the two passes that can overlap (nm_mc_emit_slab); one per device, created on first use.
struct McFork { hipStream_t side = nullptr; hipEvent_t forked = nullptr, joined = nullptr; std::mutex lock; };
static McFork* mc_fork() {
    static McFork per_device[16];
    static std::mutex create;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    McFork* f = &per_device[dev];
    std::lock_guard<std::mutex> hold(create);
    if (!f->side) {
        hipStream_t s = nullptr;
        hipEvent_t a = nullptr, b = nullptr;
        if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&a, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&b, hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;          // the passes then run one after the other
        }
        f->forked = a; f->joined = b; f->side = s;
    }
    return f;
}

extern "C" {

int64_t nm_mc_workspace_bytes(int32_t n0, int32_t n1, int32_t n2) {
    if (n0 < 2 || n1 < 2 || n2 < 2) return 0;
    return (int64_t)carve(make_dims(n0, n1, n2), nullptr, nullptr);
}

int64_t nm_mc_vertex_scratch_bytes(int64_t vertices, int64_t faces) {
    // per vertex: creating cube (8 B) + edge id (1 B); per active cube (<= faces): one 16-byte list entry
    return (int64_t)(al((size_t)vertices * 8) + al((size_t)vertices) + al((size_t)faces * sizeof(McActive)));
}

int nm_mc_count_slab(const float* d_volume, int32_t n0, int32_t n1, int32_t n2, double iso, int32_t z_global,
                     int32_t ghost_below, int32_t ghost_above, void* d_workspace, int64_t* h_vertices, int64_t* h_faces,
                     int64_t* h_ghost_vertices, int64_t* h_ghost_faces, void* stream_) {
    NM_REQUIRE(d_volume && d_workspace && h_vertices && h_faces && h_ghost_vertices && h_ghost_faces, "bad argument");
    NM_REQUIRE(n0 >= 2 && n1 >= 2 && n2 >= 2, "Input array must be at least 2x2x2.");
    NM_REQUIRE((ghost_below == 0 || ghost_below == 1) && (ghost_above == 0 || ghost_above == 1) && z_global >= 0,
               "mc slab: ghost layers are 0 or 1 planes, the global plane index is >= 0");
    NM_REQUIRE(n0 - 1 - ghost_below - ghost_above >= 1, "mc slab: the slab needs a cube layer of its own");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const McDims d = make_dims(n0, n1, n2);
    NM_REQUIRE(d.cubes < (int64_t(1) << 40), "volume too large");
    McWorkspace ws;
    carve(d, static_cast<char*>(d_workspace), &ws);
    // the layers this call classifies: everything but the ghost layer above (that one belongs to the next slab; its voxels
    // are only read when the normals of this slab's top vertices replay the cubes around them)
    McDims dc = d;
    dc.c0 = d.c0 - ghost_above;
    dc.cubes = (int64_t)dc.c0 * d.c1 * d.c2;
    const int64_t units = (int64_t)dc.c0 * d.c1 * ws.bx;
    const int c_rows = MC_MARCH0 ? dc.c1 : dc.c0, c_march = MC_MARCH0 ? dc.c0 : dc.c1;
    static int cus_of_device[64] = {};
    int dev = 0;
    NM_HIP_CHECK(hipGetDevice(&dev));
    if (dev >= 0 && dev < 64 && cus_of_device[dev] == 0) {
        hipDeviceProp_t prop;
        NM_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
        cus_of_device[dev] = prop.multiProcessorCount;
    }
    const unsigned bx = (unsigned)ws.bx, by = (unsigned)((c_rows + MC_GY - 2) / (MC_GY - 1));
    const int zrun = mc_pick_zrun((int64_t)bx * by, c_march, dev >= 0 && dev < 64 ? cus_of_device[dev] : 0);
    const dim3 bricks(bx, by, (unsigned)((c_march + zrun - 1) / zrun));
    NM_REQUIRE(bricks.y <= 65535u && bricks.z <= 65535u, "volume too large");
    float thr = (float)iso;                        // largest float <= iso (exact equivalence of the sign test)
    if ((double)thr > iso) thr = nextafterf(thr, -INFINITY);
    const bool vec = n2 % 4 == 0, xhalo = bricks.x > 1;
#define NM_STREAM(V, X) hipLaunchKernelGGL((mc_classify_stream<V, X>), bricks, dim3(MC_STREAM_THREADS), 0, stream, d_volume, dc, thr, ws.masks, zrun)
    if (vec && !xhalo) NM_STREAM(true, false);
    else if (vec) NM_STREAM(true, true);
    else if (!xhalo) NM_STREAM(false, false);
    else NM_STREAM(false, true);
#undef NM_STREAM
    const int64_t segments = (int64_t)dc.c0 * by * bx;
    hipLaunchKernelGGL(mc_classify_cut, dim3((unsigned)((segments + 3) / 4)), dim3(256), 0, stream, d_volume, dc, iso, (int)z_global,
                       ws.masks, ws.slots, ws.cap, (int)bx, (int)by, segments, ws.unit_sums, ws.entries);
    const int64_t groups = (units + 1023) / 1024;
    hipLaunchKernelGGL(mc_scan_groups, dim3((unsigned)groups), dim3(1024), 0, stream, ws.unit_sums, units, ws.group_sums);
    uint32_t* h_slot = nullptr;
    uint32_t* d_slot = nullptr;
    hipEvent_t counted = nullptr;
    NM_REQUIRE(mc_host_slot(&h_slot, &d_slot, &counted), "mc: no pinned host memory for the totals");
    const int64_t ghost_unit = ghost_below ? (int64_t)d.c1 * ws.bx : -1;   // first unit of the second layer
    hipLaunchKernelGGL(mc_scan_totals, dim3(1), dim3(1024), 0, stream, ws.group_sums, groups, ws.totals, ws.unit_sums, ghost_unit, d_slot);
    // the host waits for the totals only; the list of cut cubes is built while it allocates the outputs
    NM_HIP_CHECK(hipEventRecord(counted, stream));
    hipLaunchKernelGGL(mc_compact, dim3((unsigned)((segments + 3) / 4)), dim3(256), 0, stream, dc, ws.slots, ws.cap, ws.bx, ws.by,
                       segments, ws.unit_sums, ws.group_sums, ws.list, ws.totals, ws.list_capacity);
    NM_HIP_CHECK(hipGetLastError());
    NM_HIP_CHECK(hipEventSynchronize(counted));
    const uint32_t totals[5] = {h_slot[0], h_slot[1], h_slot[2], h_slot[3], h_slot[4]};
    *h_vertices = totals[0];
    *h_faces = totals[1];
    *h_ghost_vertices = totals[3];
    *h_ghost_faces = totals[4];
    return 0;
}

int nm_mc_count(const float* d_volume, int32_t n0, int32_t n1, int32_t n2, double iso, void* d_workspace,
                int64_t* h_vertices, int64_t* h_faces, void* stream_) {
    int64_t gv = 0, gf = 0;
    return nm_mc_count_slab(d_volume, n0, n1, n2, iso, 0, 0, 0, d_workspace, h_vertices, h_faces, &gv, &gf, stream_);
}

int nm_mc_emit_slab(const float* d_volume, int32_t n0, int32_t n1, int32_t n2, double iso, int32_t z_global,
                    int32_t ghost_below, int32_t ghost_above, void* d_workspace, void* d_vertex_scratch, int64_t vertices,
                    int64_t faces, int64_t ghost_vertices, int64_t ghost_faces, int64_t index_base, float* d_verts,
                    int32_t* d_faces, float* d_normals, float* d_values, void* stream_) {
    NM_REQUIRE(d_volume && d_workspace && d_vertex_scratch, "bad argument");
    NM_REQUIRE(ghost_vertices >= 0 && ghost_vertices <= vertices && ghost_faces >= 0 && ghost_faces <= faces, "mc slab: bad ghost counts");
    // a slab whose only vertices belong to the ghost layer below (the surface's topmost tip ends inside that layer) owns
    // nothing: its output arrays have no elements and their pointers may be null
    NM_REQUIRE(vertices == ghost_vertices || (d_verts && d_normals && d_values), "mc slab: null vertex outputs");
    NM_REQUIRE(faces == ghost_faces || d_faces, "mc slab: null face output");
    if (vertices == ghost_vertices && faces == ghost_faces) return 0;
    NM_REQUIRE(index_base + vertices < (int64_t(1) << 31) && index_base + ghost_vertices >= 0, "mc slab: vertex ids do not fit int32");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (vertices == 0) return 0;
    const McDims d = make_dims(n0, n1, n2);
    McWorkspace ws;
    carve(d, static_cast<char*>(d_workspace), &ws);
    McDims dc = d;
    dc.c0 = d.c0 - ghost_above;
    dc.cubes = (int64_t)dc.c0 * d.c1 * d.c2;
    const int64_t own_v = vertices - ghost_vertices;
    ws.vertex_cube = static_cast<int64_t*>(d_vertex_scratch);
    ws.vertex_edge = reinterpret_cast<int8_t*>(static_cast<char*>(d_vertex_scratch) + al((size_t)vertices * 8));
    ws.active = reinterpret_cast<McActive*>(static_cast<char*>(d_vertex_scratch) + al((size_t)vertices * 8) + al((size_t)vertices));
    McOut out;
    out.verts = d_verts; out.faces = d_faces; out.normals = d_normals; out.values = d_values;
    for (int a = 0; a < 3; ++a) out.edge_vertex[a] = ws.edge[a];
    McSlab sl{(int)z_global, (int)ghost_below, (uint32_t)ghost_vertices, (uint32_t)ghost_faces, index_base};
    const unsigned ablocks = (unsigned)((faces + 255) / 256);   // active cubes <= faces; surplus threads exit
    const int64_t segments = (int64_t)dc.c0 * ws.by * ws.bx;
    if ((uint64_t)faces > ws.list_capacity)     // the count call could not build the list (same test as the kernel's, on the same number)
        hipLaunchKernelGGL(mc_compact, dim3((unsigned)((segments + 3) / 4)), dim3(256), 0, stream, dc, ws.slots, ws.cap, ws.bx, ws.by,
                           segments, ws.unit_sums, ws.group_sums, ws.active, ws.totals, 0xffffffffu);
    else
        ws.active = ws.list;
    hipLaunchKernelGGL(mc_emit<false>, dim3(ablocks), dim3(256), 0, stream, d_volume, dc, iso, sl, ws.active, ws.totals, out,
                       ws.vertex_cube, ws.vertex_edge);
    // the faces pass and the attribute pass both depend on the vertex pass only, and both are latency-bound walks over
    // ~1 % of the cubes: they run side by side (fork / join through a per-device side stream; the pair of events is
    // enqueued under a lock so two host threads cannot interleave their record / wait pairs; capturable).  The longer
    // one (attributes) stays on the caller's stream: a cross-stream dependency takes 7-14 us to resolve (kernel trace,
    // round 3), which the shorter faces pass can afford and which the join, long since satisfied, then does not add.
    // (Both passes as ONE launch with alternating workgroups measured slower: 95 us against 87 for the pair.)
    const McLookup lk{ws.masks, ws.entries, ws.bx};
    McFork* fk = own_v > 0 ? mc_fork() : nullptr;
    if (fk) {
        std::lock_guard<std::mutex> hold(fk->lock);
        NM_HIP_CHECK(hipEventRecord(fk->forked, stream));
        NM_HIP_CHECK(hipStreamWaitEvent(fk->side, fk->forked, 0));
        hipLaunchKernelGGL(mc_vertex_attributes, dim3((unsigned)((own_v + 255) / 256)), dim3(256), 0, stream, d_volume, d, dc, iso,
                           lk, ws.vertex_cube, ws.vertex_edge, own_v, out);
        hipLaunchKernelGGL(mc_emit<true>, dim3(ablocks), dim3(256), 0, fk->side, d_volume, dc, iso, sl, ws.active, ws.totals, out,
                           ws.vertex_cube, ws.vertex_edge);
        NM_HIP_CHECK(hipEventRecord(fk->joined, fk->side));
        NM_HIP_CHECK(hipStreamWaitEvent(stream, fk->joined, 0));
    } else {
        hipLaunchKernelGGL(mc_emit<true>, dim3(ablocks), dim3(256), 0, stream, d_volume, dc, iso, sl, ws.active, ws.totals, out,
                           ws.vertex_cube, ws.vertex_edge);
        if (own_v > 0)
            hipLaunchKernelGGL(mc_vertex_attributes, dim3((unsigned)((own_v + 255) / 256)), dim3(256), 0, stream, d_volume, d, dc, iso,
                               lk, ws.vertex_cube, ws.vertex_edge, own_v, out);
    }
    NM_HIP_CHECK(hipGetLastError());
    return 0;
}

int nm_mc_emit(const float* d_volume, int32_t n0, int32_t n1, int32_t n2, double iso, void* d_workspace,
               void* d_vertex_scratch, int64_t vertices, int64_t faces, float* d_verts, int32_t* d_faces,
               float* d_normals, float* d_values, void* stream_) {
    return nm_mc_emit_slab(d_volume, n0, n1, n2, iso, 0, 0, 0, d_workspace, d_vertex_scratch, vertices, faces, 0, 0, 0, d_verts,
                           d_faces, d_normals, d_values, stream_);
}

}  // extern "C"
