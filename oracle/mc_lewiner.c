/*
 * CPU ORACLE (test infrastructure, NOT product code): sequential restatement of the Lewiner MC33
 * marching-cubes algorithm exactly as `skimage.measure.marching_cubes(volume, level)` runs it with its
 * defaults (method='lewiner', step_size=1, gradient_direction='descent', allow_degenerate=True, no mask).
 *
 * The reference calls it at /root/reference/src/mesh_nerf.py:79; the arithmetic lives in the third-party
 * dependency scikit-image (pinned 0.17.2 in /root/reference/requirements.txt:39), whose Cython source
 * (`_marching_cubes_lewiner_cy.pyx`) is NOT present offline -- only the compiled module of scikit-image
 * 0.18.3 under /opt/conda.  This file therefore restates the published algorithm (Lewiner, Lopes, Vieira,
 * Tavares: "Efficient implementation of Marching Cubes' cases with topological guarantees", JGT 8(2) 2003,
 * functions process_cube / test_face / test_interior) together with scikit-image's vertex cache
 * (two per-z-layer tables, 4 slots per cell: x-edge, y-edge, z-edge, centre vertex), inverse-|v| edge
 * interpolation and gradient accumulation, and is PINNED black-box against that compiled module:
 * tests/golden/make_mc_golden.py fuzzes all 256 sign patterns and random / smooth / degenerate volumes
 * and this oracle reproduces vertices, faces, normals and values bit for bit (tests/test_mc_oracle.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call into this file.
 *
 * Build: make -C oracle   ->  oracle/_build/libmc_oracle.so
 */
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "mc_luts.h"

/* scikit-image defines its "SK_EPS" as np.spacing(1.0), i.e. the DOUBLE epsilon (probed: edge
 * interpolation matches bit for bit only with 2.2e-16). */
#define SK_EPS DBL_EPSILON

typedef struct {
    int nx, ny, nz;            /* skimage naming: x = last volume axis, z = first */
    int x, y, z;
    double v[8];               /* corner values minus iso, Lewiner numbering */
    double vv[8];              /* same, indexed by dz*4 + dy*2 + dx */
    double vg[24];             /* corner "gradients" */
    double vmax;
    int index;
    int c12_done;
    double c12[3], c12g[3];
    int *layer1, *layer2, *layer; /* vertex cache */
    float *verts, *normals, *values;
    int nverts, cap_verts;
    int *faces;
    int nfaces, cap_faces;     /* counts face CORNERS */
    int lut_case, lut_config, lut_subconfig;
} Cell;

static const signed char EDGE_DX[12][2] = {{0,1},{1,1},{1,0},{0,0},{0,1},{1,1},{1,0},{0,0},{0,0},{1,1},{1,1},{0,0}};
static const signed char EDGE_DY[12][2] = {{0,0},{0,1},{1,1},{1,0},{0,0},{0,1},{1,1},{1,0},{0,0},{0,0},{1,1},{1,1}};
static const signed char EDGE_DZ[12][2] = {{0,0},{0,0},{0,0},{0,0},{1,1},{1,1},{1,1},{1,1},{0,1},{0,1},{0,1},{0,1}};

static int add_vertex(Cell* c, float x, float y, float z) {
    if (c->nverts == c->cap_verts) {
        c->cap_verts *= 2;
        c->verts = (float*)realloc(c->verts, sizeof(float) * 3 * c->cap_verts);
        c->normals = (float*)realloc(c->normals, sizeof(float) * 3 * c->cap_verts);
        c->values = (float*)realloc(c->values, sizeof(float) * c->cap_verts);
    }
    const int i = c->nverts++;
    c->verts[3 * i] = x; c->verts[3 * i + 1] = y; c->verts[3 * i + 2] = z;
    c->normals[3 * i] = c->normals[3 * i + 1] = c->normals[3 * i + 2] = 0.0f;
    c->values[i] = 0.0f;
    return i;
}

static void add_gradient(Cell* c, int vi, float gx, float gy, float gz) {
    c->normals[3 * vi] += gx; c->normals[3 * vi + 1] += gy; c->normals[3 * vi + 2] += gz;
}

static void add_gradient_from_index(Cell* c, int vi, int corner, float strength) {   /* strength is a C float in skimage */
    add_gradient(c, vi, (float)(c->vg[3 * corner] * strength), (float)(c->vg[3 * corner + 1] * strength),
                 (float)(c->vg[3 * corner + 2] * strength));
}

static void add_face(Cell* c, int vi) {
    if (c->nfaces == c->cap_faces) {
        c->cap_faces *= 2;
        c->faces = (int*)realloc(c->faces, sizeof(int) * c->cap_faces);
    }
    c->faces[c->nfaces++] = vi;
    if (c->vmax > c->values[vi]) c->values[vi] = (float)c->vmax;
}

static void prepare(Cell* c) {
    const double* v = c->v;
    c->vv[0] = v[0]; c->vv[1] = v[1]; c->vv[2] = v[3]; c->vv[3] = v[2];
    c->vv[4] = v[4]; c->vv[5] = v[5]; c->vv[6] = v[7]; c->vv[7] = v[6];
    double lo = 0.0, hi = 0.0;
    for (int i = 0; i < 8; ++i) {
        if (c->vv[i] > hi) hi = c->vv[i];
        if (c->vv[i] < lo) lo = c->vv[i];
    }
    c->vmax = hi - lo;
    double* g = c->vg;
    g[0] = v[0] - v[1];  g[1] = v[0] - v[3];  g[2] = v[0] - v[4];
    g[3] = v[0] - v[1];  g[4] = v[1] - v[2];  g[5] = v[1] - v[5];
    g[6] = v[3] - v[2];  g[7] = v[1] - v[2];  g[8] = v[2] - v[6];
    g[9] = v[3] - v[2];  g[10] = v[0] - v[3]; g[11] = v[3] - v[7];
    g[12] = v[4] - v[5]; g[13] = v[4] - v[7]; g[14] = v[0] - v[4];
    g[15] = v[4] - v[5]; g[16] = v[5] - v[6]; g[17] = v[1] - v[5];
    g[18] = v[7] - v[6]; g[19] = v[5] - v[6]; g[20] = v[2] - v[6];
    g[21] = v[7] - v[6]; g[22] = v[4] - v[7]; g[23] = v[3] - v[7];
}

static void centre_vertex(Cell* c) {
    static const double CX[8] = {0, 1, 1, 0, 0, 1, 1, 0}, CY[8] = {0, 0, 1, 1, 0, 0, 1, 1}, CZ[8] = {0, 0, 0, 0, 1, 1, 1, 1};
    double w[8], fx = 0, fy = 0, fz = 0, ff = 0;
    for (int i = 0; i < 8; ++i) w[i] = 1.0 / (SK_EPS + fabs(c->v[i]));
    for (int i = 0; i < 8; ++i) { fx += CX[i] * w[i]; fy += CY[i] * w[i]; fz += CZ[i] * w[i]; ff += w[i]; }
    c->c12[0] = c->x + fx / ff; c->c12[1] = c->y + fy / ff; c->c12[2] = c->z + fz / ff;
    /* scikit-image quirk (black-box probed): the centre vertex's gradient comes out as
     * (sum w*g_z, sum w*g_y, 0) -- its z-sum lands in the x slot and the z slot stays zero. */
    double gs[3];
    for (int a = 0; a < 3; ++a) {
        double s = 0;
        for (int i = 0; i < 8; ++i) s += w[i] * c->vg[3 * i + a];
        gs[a] = s;
    }
    c->c12g[0] = gs[2]; c->c12g[1] = gs[1]; c->c12g[2] = 0.0;
    c->c12_done = 1;
}

static int cache_slot(Cell* c, int vi) {
    int i = c->nx * c->y + c->x, j = 0;
    if (vi < 8) {
        if (vi < 4) c->layer = c->layer1; else { vi -= 4; c->layer = c->layer2; }
        if (vi == 1) { i += 1; j = 1; }
        else if (vi == 2) { i += c->nx; }
        else if (vi == 3) { j = 1; }
    } else if (vi < 12) {
        c->layer = c->layer1; j = 2;
        if (vi == 9) i += 1;
        else if (vi == 10) i += c->nx + 1;
        else if (vi == 11) i += c->nx;
    } else {
        c->layer = c->layer1; j = 3;
    }
    return 4 * i + j;
}

static void face_from_edge(Cell* c, int vi) {
    const int slot = cache_slot(c, vi);
    int idx = c->layer[slot];
    if (vi == 12) {
        if (!c->c12_done) centre_vertex(c);
        if (idx >= 0) {   /* every further reference adds the centre gradient again (probed) */
            add_face(c, idx);
            add_gradient(c, idx, (float)c->c12g[0], (float)c->c12g[1], (float)c->c12g[2]);
            return;
        }
        idx = add_vertex(c, (float)c->c12[0], (float)c->c12[1], (float)c->c12[2]);
        c->layer[slot] = idx;
        add_face(c, idx);
        add_gradient(c, idx, (float)c->c12g[0], (float)c->c12g[1], (float)c->c12g[2]);
        return;
    }
    const int dx1 = EDGE_DX[vi][0], dx2 = EDGE_DX[vi][1], dy1 = EDGE_DY[vi][0], dy2 = EDGE_DY[vi][1];
    const int dz1 = EDGE_DZ[vi][0], dz2 = EDGE_DZ[vi][1];
    const int i1 = dz1 * 4 + dy1 * 2 + dx1, i2 = dz2 * 4 + dy2 * 2 + dx2;
    const double w1 = 1.0 / (SK_EPS + fabs(c->vv[i1])), w2 = 1.0 / (SK_EPS + fabs(c->vv[i2]));
    if (idx < 0) {
        double fx = 0, fy = 0, fz = 0, ff = 0;
        fx += dx1 * w1; fy += dy1 * w1; fz += dz1 * w1; ff += w1;
        fx += dx2 * w2; fy += dy2 * w2; fz += dz2 * w2; ff += w2;
        idx = add_vertex(c, (float)(c->x + fx / ff), (float)(c->y + fy / ff), (float)(c->z + fz / ff));
        c->layer[slot] = idx;
    }
    add_face(c, idx);
    add_gradient_from_index(c, idx, i1, w1);
    add_gradient_from_index(c, idx, i2, w2);
}

static void add_triangles(Cell* c, int offset, int nt) {
    prepare(c);
    for (int i = 0; i < 3 * nt; ++i) face_from_edge(c, MC_LUT[offset + i]);
}

static int test_face(const Cell* c, int face) {
    const double* v = c->v;
    double A, B, C, D;
    switch (abs(face)) {
        case 1: A = v[0]; B = v[4]; C = v[5]; D = v[1]; break;
        case 2: A = v[1]; B = v[5]; C = v[6]; D = v[2]; break;
        case 3: A = v[2]; B = v[6]; C = v[7]; D = v[3]; break;
        case 4: A = v[3]; B = v[7]; C = v[4]; D = v[0]; break;
        case 5: A = v[0]; B = v[3]; C = v[2]; D = v[1]; break;
        default: A = v[4]; B = v[7]; C = v[6]; D = v[5]; break;
    }
    const double acbd = A * C - B * D;
    if (acbd > -SK_EPS && acbd < SK_EPS) return face >= 0;
    return face * A * acbd >= 0;
}

static int test_interior(const Cell* c, int s) {
    const double* v = c->v;
    double t, At = 0, Bt = 0, Ct = 0, Dt = 0, a, b;
    int test = 0, edge = -1;
    const int kase = c->lut_case, cfg = c->lut_config;
    if (kase == 4 || kase == 10) {
        a = (v[4] - v[0]) * (v[6] - v[2]) - (v[7] - v[3]) * (v[5] - v[1]);
        b = v[2] * (v[4] - v[0]) + v[0] * (v[6] - v[2]) - v[1] * (v[7] - v[3]) - v[3] * (v[5] - v[1]);
        t = -b / (2 * a + SK_EPS);   /* scikit-image guards its divisions with +eps (probed: a == b == 0) */
        if (t < 0 || t > 1) return s > 0;
        At = v[0] + (v[4] - v[0]) * t;
        Bt = v[3] + (v[7] - v[3]) * t;
        Ct = v[2] + (v[6] - v[2]) * t;
        Dt = v[1] + (v[5] - v[1]) * t;
    } else {
        if (kase == 6) edge = MC_LUT[MC_TEST6_OFF + cfg * MC_TEST6_ROW + 2];
        else if (kase == 7) edge = MC_LUT[MC_TEST7_OFF + cfg * MC_TEST7_ROW + 4];
        else if (kase == 12) edge = MC_LUT[MC_TEST12_OFF + cfg * MC_TEST12_ROW + 3];
        else edge = MC_LUT[MC_TILING13_5_1_OFF + (cfg * MC_TILING13_5_1_SUB + c->lut_subconfig) * MC_TILING13_5_1_ROW];
        /* reference edge e = (p, q); the three parallel edges in order B, C, D */
        static const signed char E[12][8] = {
            {0, 1, 3, 2, 7, 6, 4, 5}, {1, 2, 0, 3, 4, 7, 5, 6}, {2, 3, 1, 0, 5, 4, 6, 7}, {3, 0, 2, 1, 6, 5, 7, 4},
            {4, 5, 7, 6, 3, 2, 0, 1}, {5, 6, 4, 7, 0, 3, 1, 2}, {6, 7, 5, 4, 1, 0, 2, 3}, {7, 4, 6, 5, 2, 1, 3, 0},
            {0, 4, 3, 7, 2, 6, 1, 5}, {1, 5, 0, 4, 3, 7, 2, 6}, {2, 6, 1, 5, 0, 4, 3, 7}, {3, 7, 2, 6, 1, 5, 0, 4}};
        const signed char* e = E[edge];
        t = v[e[0]] / (v[e[0]] - v[e[1]] + SK_EPS);
        At = 0;
        Bt = v[e[2]] + (v[e[3]] - v[e[2]]) * t;
        Ct = v[e[4]] + (v[e[5]] - v[e[4]]) * t;
        Dt = v[e[6]] + (v[e[7]] - v[e[6]]) * t;
    }
    if (At >= 0) test += 1;
    if (Bt >= 0) test += 2;
    if (Ct >= 0) test += 4;
    if (Dt >= 0) test += 8;
    switch (test) {
        case 0: case 1: case 2: case 3: case 4: case 6: case 8: case 9: case 12: return s > 0;
        /* Lewiner's C++ falls through to `return s < 0` when the saddle test below fails; the scikit-image
         * port returns 0 there (black-box probed on 4000 case-4 cubes: for s < 0 the two differ). */
        case 5: if (At * Ct - Bt * Dt < SK_EPS) return s > 0; return 0;
        case 10: if (At * Ct - Bt * Dt >= SK_EPS) return s > 0; return 0;
        default: return s < 0;   /* 7, 11, 13, 14, 15 */
    }
}

#define ROW(NAME, cfg) (MC_##NAME##_OFF + (cfg) * MC_##NAME##_ROW)
#define ROW3(NAME, cfg, sub) (MC_##NAME##_OFF + ((cfg) * MC_##NAME##_SUB + (sub)) * MC_##NAME##_ROW)
#define T1(NAME, cfg) MC_LUT[MC_##NAME##_OFF + (cfg)]
#define T2(NAME, cfg, k) MC_LUT[MC_##NAME##_OFF + (cfg) * MC_##NAME##_ROW + (k)]

/* Lewiner's process_cube dispatcher: picks the tiling (offset into MC_LUT, triangle count). */
static void big_switch(Cell* c, int kase, int cfg) {
    int sub = 0;
    c->lut_case = kase; c->lut_config = cfg; c->lut_subconfig = 0;
    switch (kase) {
        case 1: add_triangles(c, ROW(TILING1, cfg), 1); break;
        case 2: add_triangles(c, ROW(TILING2, cfg), 2); break;
        case 3:
            if (test_face(c, T1(TEST3, cfg))) add_triangles(c, ROW(TILING3_2, cfg), 4);
            else add_triangles(c, ROW(TILING3_1, cfg), 2);
            break;
        case 4:
            if (test_interior(c, T1(TEST4, cfg))) add_triangles(c, ROW(TILING4_1, cfg), 2);
            else add_triangles(c, ROW(TILING4_2, cfg), 6);
            break;
        case 5: add_triangles(c, ROW(TILING5, cfg), 3); break;
        case 6:
            if (test_face(c, T2(TEST6, cfg, 0))) add_triangles(c, ROW(TILING6_2, cfg), 5);
            else if (test_interior(c, T2(TEST6, cfg, 1))) add_triangles(c, ROW(TILING6_1_1, cfg), 3);
            else add_triangles(c, ROW(TILING6_1_2, cfg), 9);
            break;
        case 7:
            if (test_face(c, T2(TEST7, cfg, 0))) sub += 1;
            if (test_face(c, T2(TEST7, cfg, 1))) sub += 2;
            if (test_face(c, T2(TEST7, cfg, 2))) sub += 4;
            switch (sub) {
                case 0: add_triangles(c, ROW(TILING7_1, cfg), 3); break;
                case 1: add_triangles(c, ROW3(TILING7_2, cfg, 0), 5); break;
                case 2: add_triangles(c, ROW3(TILING7_2, cfg, 1), 5); break;
                case 3: add_triangles(c, ROW3(TILING7_3, cfg, 0), 9); break;
                case 4: add_triangles(c, ROW3(TILING7_2, cfg, 2), 5); break;
                case 5: add_triangles(c, ROW3(TILING7_3, cfg, 1), 9); break;
                case 6: add_triangles(c, ROW3(TILING7_3, cfg, 2), 9); break;
                default:
                    if (test_interior(c, T2(TEST7, cfg, 3))) add_triangles(c, ROW(TILING7_4_2, cfg), 9);
                    else add_triangles(c, ROW(TILING7_4_1, cfg), 5);
            }
            break;
        case 8: add_triangles(c, ROW(TILING8, cfg), 2); break;
        case 9: add_triangles(c, ROW(TILING9, cfg), 4); break;
        case 10:
            if (test_face(c, T2(TEST10, cfg, 0))) {
                if (test_face(c, T2(TEST10, cfg, 1))) add_triangles(c, ROW(TILING10_1_1_, cfg), 4);
                else add_triangles(c, ROW(TILING10_2, cfg), 8);
            } else {
                if (test_face(c, T2(TEST10, cfg, 1))) add_triangles(c, ROW(TILING10_2_, cfg), 8);
                else if (test_interior(c, T2(TEST10, cfg, 2))) add_triangles(c, ROW(TILING10_1_1, cfg), 4);
                else add_triangles(c, ROW(TILING10_1_2, cfg), 8);
            }
            break;
        case 11: add_triangles(c, ROW(TILING11, cfg), 4); break;
        case 12:
            if (test_face(c, T2(TEST12, cfg, 0))) {
                if (test_face(c, T2(TEST12, cfg, 1))) add_triangles(c, ROW(TILING12_1_1_, cfg), 4);
                else add_triangles(c, ROW(TILING12_2, cfg), 8);
            } else {
                if (test_face(c, T2(TEST12, cfg, 1))) add_triangles(c, ROW(TILING12_2_, cfg), 8);
                else if (test_interior(c, T2(TEST12, cfg, 2))) add_triangles(c, ROW(TILING12_1_1, cfg), 4);
                else add_triangles(c, ROW(TILING12_1_2, cfg), 8);
            }
            break;
        case 13: {
            for (int k = 0; k < 6; ++k)
                if (test_face(c, T2(TEST13, cfg, k))) sub += 1 << k;
            const int sc = MC_LUT[MC_SUBCONFIG13_OFF + sub];
            if (sc == 0) add_triangles(c, ROW(TILING13_1, cfg), 4);
            else if (sc <= 6) add_triangles(c, ROW3(TILING13_2, cfg, sc - 1), 6);
            else if (sc <= 18) add_triangles(c, ROW3(TILING13_3, cfg, sc - 7), 10);
            else if (sc <= 22) add_triangles(c, ROW3(TILING13_4, cfg, sc - 19), 12);
            else if (sc <= 26) {
                c->lut_subconfig = sc - 23;
                if (test_interior(c, T2(TEST13, cfg, 6))) add_triangles(c, ROW3(TILING13_5_1, cfg, sc - 23), 6);
                else add_triangles(c, ROW3(TILING13_5_2, cfg, sc - 23), 10);
            } else if (sc <= 38) add_triangles(c, ROW3(TILING13_3_, cfg, sc - 27), 10);
            else if (sc <= 44) add_triangles(c, ROW3(TILING13_2_, cfg, sc - 39), 6);
            else if (sc == 45) add_triangles(c, ROW(TILING13_1_, cfg), 4);
            break;
        }
        case 14: add_triangles(c, ROW(TILING14, cfg), 4); break;
        default: break;
    }
}

/* volume: (n0, n1, n2) C-contiguous fp32 (axis 2 fastest).  Outputs are malloc'ed; in the layout
 * skimage.measure.marching_cubes returns: vertices / normals columns in (axis0, axis1, axis2) order,
 * faces with the corner order reversed ('descent'), normals normalised.  Returns 0, or 1 when no
 * surface is found (skimage raises RuntimeError). */
int mc_lewiner(const float* vol, int n0, int n1, int n2, double iso, float** out_verts, int* out_nverts,
               int** out_faces, int* out_nfaces, float** out_normals, float** out_values) {
    Cell c;
    memset(&c, 0, sizeof(c));
    c.nx = n2; c.ny = n1; c.nz = n0;
    const size_t slots = (size_t)c.nx * c.ny * 4;
    c.layer1 = (int*)malloc(sizeof(int) * slots);
    c.layer2 = (int*)malloc(sizeof(int) * slots);
    for (size_t i = 0; i < slots; ++i) c.layer1[i] = c.layer2[i] = -1;
    c.cap_verts = 1024; c.cap_faces = 4096;
    c.verts = (float*)malloc(sizeof(float) * 3 * c.cap_verts);
    c.normals = (float*)malloc(sizeof(float) * 3 * c.cap_verts);
    c.values = (float*)malloc(sizeof(float) * c.cap_verts);
    c.faces = (int*)malloc(sizeof(int) * c.cap_faces);
#define VOL(z, y, x) ((double)vol[((size_t)(z) * n1 + (y)) * n2 + (x)])
    for (int z = 0; z < c.nz - 1; ++z) {
        int* tmp = c.layer1; c.layer1 = c.layer2; c.layer2 = tmp;     /* new_z_value */
        for (size_t i = 0; i < slots; ++i) c.layer2[i] = -1;
        for (int y = 0; y < c.ny - 1; ++y)
            for (int x = 0; x < c.nx - 1; ++x) {
                c.x = x; c.y = y; c.z = z;
                c.v[0] = VOL(z, y, x) - iso;         c.v[1] = VOL(z, y, x + 1) - iso;
                c.v[2] = VOL(z, y + 1, x + 1) - iso; c.v[3] = VOL(z, y + 1, x) - iso;
                c.v[4] = VOL(z + 1, y, x) - iso;         c.v[5] = VOL(z + 1, y, x + 1) - iso;
                c.v[6] = VOL(z + 1, y + 1, x + 1) - iso; c.v[7] = VOL(z + 1, y + 1, x) - iso;
                int index = 0;
                for (int k = 0; k < 8; ++k)
                    if (c.v[k] > 0.0) index |= 1 << k;
                c.index = index; c.c12_done = 0;
                const int kase = MC_LUT[MC_CASES_OFF + 2 * index];
                if (kase > 0) big_switch(&c, kase, MC_LUT[MC_CASES_OFF + 2 * index + 1]);
            }
    }
    free(c.layer1); free(c.layer2);
    /* get_normals: normalise; wrapper: flip columns, reverse faces */
    for (int i = 0; i < c.nverts; ++i) {
        float* n = c.normals + 3 * i;
        const double len = sqrt((double)n[0] * n[0] + (double)n[1] * n[1] + (double)n[2] * n[2]);
        if (len > 0.0) { n[0] = (float)(n[0] / len); n[1] = (float)(n[1] / len); n[2] = (float)(n[2] / len); }
        float t = n[0]; n[0] = n[2]; n[2] = t;
        float* p = c.verts + 3 * i;
        t = p[0]; p[0] = p[2]; p[2] = t;
    }
    for (int f = 0; f + 2 < c.nfaces; f += 3) { int t = c.faces[f]; c.faces[f] = c.faces[f + 2]; c.faces[f + 2] = t; }
    *out_verts = c.verts; *out_nverts = c.nverts; *out_faces = c.faces; *out_nfaces = c.nfaces / 3;
    *out_normals = c.normals; *out_values = c.values;
    return c.nverts == 0;
}

void mc_free(void* p) { free(p); }
