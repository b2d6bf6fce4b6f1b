/* Language-neutral use of the C ABI (include/nerfmeshes_hip.h): no Python, no torch.
 * Builds a 4x64 FlexibleNeRFModel from weights read from a raw fp32 file, evaluates n points with
 * nm_mlp_sample_points and writes the (n,4) result to a raw fp32 file; then runs the training entry points
 * (nm_mlp_forward_train + nm_mlp_backward with dL/d(radiance) = 1) on the same points as one-sample rays and appends
 * the radiance again, delta at layer1's output (n,H) and the head deltas (n,4).  Then the two entry points INTEGRATION.md B
 * tells a maintainer to bind first: the one-call renderer nm_render_rays (R = 256 rays from the given points / directions,
 * 16 coarse + 24 fine samples, the same network as coarse and fine) -> fine rgb_map (R,3) and acc_map (R,), and mesh
 * extraction -- nm_mlp_grid_query (density only, 24^3) -> nm_mc_count -> nm_mc_emit at iso = the grid's mean -> the grid,
 * iso, V, F, vertices, faces, normals, values.  The pytest driver (tests/test_gpu_cabi_c.py) generates the inputs and
 * checks everything against the CPU oracle.
 *
 *   gcc -D__HIP_PLATFORM_AMD__ tests/cabi_smoke.c -Iinclude -I/opt/rocm/include -Lnerfmeshes_amd/csrc -lnerfmeshes_hip \
 *       -L/opt/rocm/lib -lamdhip64 -o cabi_smoke
 *   ./cabi_smoke weights.bin points.bin n out.bin
 */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>

#include "nerfmeshes_hip.h"

#define H 64
#define L 4
#define FX 6
#define FD 4
#define DX (3 + 6 * FX)
#define DD (3 + 6 * FD)

static float* take(float** cursor, size_t n) { float* p = *cursor; *cursor += n; return p; }

int main(int argc, char** argv) {
    if (argc != 5) { fprintf(stderr, "usage: %s weights.bin points.bin n out.bin\n", argv[0]); return 2; }
    const long n = atol(argv[3]);
    const size_t nw = (size_t)H * DX + H + (size_t)(L - 1) * (H * H + H) + (size_t)(H / 2) * (H + DD) + H / 2 + H + 1 +
                      3 * (H / 2) + 3 + (size_t)H * H + H + FX + FD;
    float* w = (float*)malloc(nw * sizeof(float));
    FILE* f = fopen(argv[1], "rb");
    if (!f || fread(w, sizeof(float), nw, f) != nw) { fprintf(stderr, "cannot read %zu weights\n", nw); return 3; }
    fclose(f);
    float* pts = (float*)malloc((size_t)n * 6 * sizeof(float));   /* n points then n directions */
    f = fopen(argv[2], "rb");
    if (!f || fread(pts, sizeof(float), (size_t)n * 6, f) != (size_t)n * 6) { fprintf(stderr, "cannot read points\n"); return 3; }
    fclose(f);

    /* same order as the Python side writes them (tests/test_gpu_cabi_c.py) */
    float* c = w;
    nm_mlp_weights mw;
    const float* xs_w[L - 1];
    const float* xs_b[L - 1];
    mw.layer1_w = take(&c, (size_t)H * DX); mw.layer1_b = take(&c, H);
    for (int i = 0; i < L - 1; ++i) { xs_w[i] = take(&c, (size_t)H * H); xs_b[i] = take(&c, H); }
    mw.layers_xyz_w = xs_w; mw.layers_xyz_b = xs_b;
    mw.layers_dir0_w = take(&c, (size_t)(H / 2) * (H + DD)); mw.layers_dir0_b = take(&c, H / 2);
    mw.fc_alpha_w = take(&c, H); mw.fc_alpha_b = take(&c, 1);
    mw.fc_rgb_w = take(&c, 3 * (H / 2)); mw.fc_rgb_b = take(&c, 3);
    mw.fc_feat_w = take(&c, (size_t)H * H); mw.fc_feat_b = take(&c, H);
    mw.freq_xyz = take(&c, FX); mw.freq_dir = take(&c, FD);

    nm_mlp_desc d = {L, H, 4, FX, FD, 1, 1, 1};
    nm_mlp* mlp = NULL;
    if (nm_mlp_create(&d, &mw, 0, &mlp)) { fprintf(stderr, "nm_mlp_create: %s\n", nm_last_error()); return 4; }

    float *d_pts, *d_dirs, *d_out;
    hipMalloc((void**)&d_pts, (size_t)n * 12); hipMalloc((void**)&d_dirs, (size_t)n * 12); hipMalloc((void**)&d_out, (size_t)n * 16);
    hipMemcpy(d_pts, pts, (size_t)n * 12, hipMemcpyHostToDevice);
    hipMemcpy(d_dirs, pts + 3 * n, (size_t)n * 12, hipMemcpyHostToDevice);
    if (nm_mlp_sample_points(mlp, d_pts, d_dirs, n, d_out, NULL)) { fprintf(stderr, "sample_points: %s\n", nm_last_error()); return 5; }
    float* out = (float*)malloc((size_t)n * 16);
    if (hipMemcpy(out, d_out, (size_t)n * 16, hipMemcpyDeviceToHost) != hipSuccess) { fprintf(stderr, "copy back failed\n"); return 6; }
    f = fopen(argv[4], "wb");
    fwrite(out, sizeof(float), (size_t)n * 4, f);

    /* ---- training ABI: every point is a one-sample ray (origin = point, t = 0) */
    const size_t tiles = ((size_t)n + 15) / 16;
    nm_mlp_tape tape = {0};       /* the optional members (d_enc_xyz / d_enc_dir, ABI v5) stay NULL: not wanted here */
    nm_mlp_deltas dl;
    float *d_t, *d_grad, *d_rad2;
    hipMalloc((void**)&tape.d_h, (size_t)L * n * H * 4); hipMalloc((void**)&tape.d_feat, (size_t)n * H * 4);
    hipMalloc((void**)&tape.d_v, (size_t)n * (H / 2) * 4);
    hipMalloc((void**)&tape.d_mask_h, (size_t)L * tiles * 64 * 8); hipMalloc((void**)&tape.d_mask_v, tiles * 64 * 8);
    hipMalloc((void**)&dl.d_h, (size_t)L * n * H * 4); hipMalloc((void**)&dl.d_feat, (size_t)n * H * 4);
    hipMalloc((void**)&dl.d_v, (size_t)n * (H / 2) * 4); hipMalloc((void**)&dl.d_last, (size_t)n * 16);
    hipMalloc((void**)&d_t, (size_t)n * 4); hipMalloc((void**)&d_grad, (size_t)n * 16); hipMalloc((void**)&d_rad2, (size_t)n * 16);
    hipMemset(d_t, 0, (size_t)n * 4);
    float* ones = (float*)malloc((size_t)n * 16);
    for (long i = 0; i < 4 * n; ++i) ones[i] = 1.0f;
    hipMemcpy(d_grad, ones, (size_t)n * 16, hipMemcpyHostToDevice);
    if (nm_mlp_forward_train(mlp, d_pts, 1, d_dirs, d_t, n, 1, &tape, d_rad2, NULL)) { fprintf(stderr, "forward_train: %s\n", nm_last_error()); return 7; }
    if (nm_mlp_backward(mlp, n, &tape, d_rad2, d_grad, &dl, NULL)) { fprintf(stderr, "backward: %s\n", nm_last_error()); return 8; }
    float* buf = (float*)malloc((size_t)n * H * 4);
    if (hipMemcpy(out, d_rad2, (size_t)n * 16, hipMemcpyDeviceToHost) != hipSuccess) return 9;
    fwrite(out, sizeof(float), (size_t)n * 4, f);
    if (hipMemcpy(buf, dl.d_h, (size_t)n * H * 4, hipMemcpyDeviceToHost) != hipSuccess) return 9;
    fwrite(buf, sizeof(float), (size_t)n * H, f);
    if (hipMemcpy(out, dl.d_last, (size_t)n * 16, hipMemcpyDeviceToHost) != hipSuccess) return 9;
    fwrite(out, sizeof(float), (size_t)n * 4, f);

    /* ---- one weight gradient from plain C: grad(layers_xyz[0].weight) = d_h[1]^T @ tape.h[0], n = 1000 rows (NOT a multiple
     *      of 16: nm_weight_grad_ex takes any row count) + its bias gradient, and the fc_alpha / fc_rgb head rows */
    {
        const int cus = nm_mlp_num_cus(mlp);
        void* d_ws;
        float *d_dw, *d_db, *d_hw;
        hipMalloc(&d_ws, (size_t)nm_weight_grad_workspace_bytes_ex(H, H, H, H, cus));
        hipMalloc((void**)&d_dw, (size_t)H * H * 4); hipMalloc((void**)&d_db, (size_t)H * 4); hipMalloc((void**)&d_hw, (size_t)4 * H * 4);
        if (nm_weight_grad_ex(cus, dl.d_h + (size_t)n * H, H, H, tape.d_h, H, H, n, d_ws, d_dw, H, 0, d_db, NULL)) {
            fprintf(stderr, "weight_grad_ex: %s\n", nm_last_error()); return 14;
        }
        float* g = (float*)malloc((size_t)(H * H + H + 4 * H) * 4);
        if (hipMemcpy(g, d_dw, (size_t)H * H * 4, hipMemcpyDeviceToHost) != hipSuccess) return 9;
        if (hipMemcpy(g + H * H, d_db, (size_t)H * 4, hipMemcpyDeviceToHost) != hipSuccess) return 9;
        void* d_hws;
        hipMalloc(&d_hws, (size_t)nm_head_grad_workspace_bytes_ex(H));
        if (nm_head_grad_ex(dl.d_last, tape.d_h + (size_t)(L - 1) * n * H, H, H, n, d_hws, d_hw, NULL, NULL)) {
            fprintf(stderr, "head_grad_ex: %s\n", nm_last_error()); return 15;
        }
        if (hipMemcpy(g + H * H + H, d_hw, (size_t)4 * H * 4, hipMemcpyDeviceToHost) != hipSuccess) return 9;
        fwrite(g, sizeof(float), (size_t)(H * H + H + 4 * H), f);
    }

    /* ---- nm_render_rays: NeRFModel.forward in one call (src/models/model_nerf.py:37-78) */
    {
        enum { R = 256, NC = 16, NF = 24 };
        if (n < R) { fprintf(stderr, "need at least %d points\n", R); return 10; }
        float u_c[NC], u_f[NF], bounds[2] = {0.5f, 3.0f};
        for (int i = 0; i < NC; ++i) u_c[i] = (float)i / (float)(NC - 1);
        for (int i = 0; i < NF; ++i) u_f[i] = (float)i / (float)(NF - 1);
        float *d_uc, *d_uf, *d_bounds, *d_rgb, *d_depth, *d_w, *d_mw, *d_acc, *d_disp;
        void* d_ws;
        hipMalloc((void**)&d_uc, sizeof u_c); hipMalloc((void**)&d_uf, sizeof u_f); hipMalloc((void**)&d_bounds, sizeof bounds);
        hipMemcpy(d_uc, u_c, sizeof u_c, hipMemcpyHostToDevice); hipMemcpy(d_uf, u_f, sizeof u_f, hipMemcpyHostToDevice);
        hipMemcpy(d_bounds, bounds, sizeof bounds, hipMemcpyHostToDevice);
        hipMalloc(&d_ws, (size_t)nm_render_workspace_bytes(R, NC, NF));
        hipMalloc((void**)&d_rgb, R * 12); hipMalloc((void**)&d_depth, R * 4); hipMalloc((void**)&d_acc, R * 4); hipMalloc((void**)&d_disp, R * 4);
        hipMalloc((void**)&d_w, (size_t)R * (NC + NF) * 4); hipMalloc((void**)&d_mw, (size_t)R * (NC + NF) * 4);
        nm_render_cfg cfg = {NC, NF, 0, 0, 0, 1e-5f};
        nm_bundle_out fine_out = {d_rgb, d_depth, d_w, d_mw, d_acc, d_disp};
        nm_bundle_out coarse_out;                 /* the coarse bundle is mandatory: its own (R,.) arrays */
        hipMalloc((void**)&coarse_out.d_rgb_map, R * 12); hipMalloc((void**)&coarse_out.d_depth_map, R * 4);
        hipMalloc((void**)&coarse_out.d_acc_map, R * 4); hipMalloc((void**)&coarse_out.d_disp_map, R * 4);
        hipMalloc((void**)&coarse_out.d_weights, (size_t)R * NC * 4); hipMalloc((void**)&coarse_out.d_mask_weights, (size_t)R * NC * 4);
        if (nm_render_rays(mlp, mlp, &cfg, d_pts, 1, d_dirs, d_bounds, d_bounds + 1, 0, d_uc, d_uf, R, d_ws, &coarse_out, &fine_out, NULL)) {
            fprintf(stderr, "render_rays: %s\n", nm_last_error()); return 10;
        }
        float img[R * 4];
        if (hipMemcpy(img, d_rgb, R * 12, hipMemcpyDeviceToHost) != hipSuccess) return 9;
        if (hipMemcpy(img + 3 * R, d_acc, R * 4, hipMemcpyDeviceToHost) != hipSuccess) return 9;
        fwrite(img, sizeof(float), R * 4, f);
    }

    /* ---- mesh extraction: nm_mlp_grid_query -> nm_mc_count -> nm_mc_emit (src/mesh_nerf.py:27-53, 73-79) */
    {
        enum { G = 24 };
        float ax[G];
        for (int i = 0; i < G; ++i) ax[i] = -1.2f + 2.4f * (float)i / (float)(G - 1);
        float *d_ax, *d_grid;
        hipMalloc((void**)&d_ax, sizeof ax); hipMalloc((void**)&d_grid, (size_t)G * G * G * 4);
        hipMemcpy(d_ax, ax, sizeof ax, hipMemcpyHostToDevice);
        if (nm_mlp_grid_query(mlp, d_ax, d_ax, d_ax, G, G, G, 0, (int64_t)G * G * G, 1, d_grid, NULL)) { fprintf(stderr, "grid_query: %s\n", nm_last_error()); return 11; }
        float* grid = (float*)malloc((size_t)G * G * G * 4);
        if (hipMemcpy(grid, d_grid, (size_t)G * G * G * 4, hipMemcpyDeviceToHost) != hipSuccess) return 9;
        double sum = 0;
        for (long i = 0; i < (long)G * G * G; ++i) sum += grid[i];
        const float iso = (float)(sum / ((double)G * G * G));
        void *d_mcws, *d_scratch;
        hipMalloc(&d_mcws, (size_t)nm_mc_workspace_bytes(G, G, G));
        int64_t nv = 0, nf = 0;
        if (nm_mc_count(d_grid, G, G, G, (double)iso, d_mcws, &nv, &nf, NULL)) { fprintf(stderr, "mc_count: %s\n", nm_last_error()); return 12; }
        float head[3] = {iso, (float)nv, (float)nf};
        fwrite(grid, sizeof(float), (size_t)G * G * G, f);
        fwrite(head, sizeof(float), 3, f);
        if (nv > 0) {
            float *d_v, *d_nrm, *d_val;
            int32_t* d_f;
            hipMalloc(&d_scratch, (size_t)nm_mc_vertex_scratch_bytes(nv, nf) + 256);
            hipMalloc((void**)&d_v, (size_t)nv * 12); hipMalloc((void**)&d_nrm, (size_t)nv * 12); hipMalloc((void**)&d_val, (size_t)nv * 4);
            hipMalloc((void**)&d_f, (size_t)nf * 12);
            if (nm_mc_emit(d_grid, G, G, G, (double)iso, d_mcws, d_scratch, nv, nf, d_v, d_f, d_nrm, d_val, NULL)) { fprintf(stderr, "mc_emit: %s\n", nm_last_error()); return 13; }
            const size_t big = (size_t)(nv > nf ? nv : nf) * 12;
            void* host = malloc(big);
            if (hipMemcpy(host, d_v, (size_t)nv * 12, hipMemcpyDeviceToHost) != hipSuccess) return 9;
            fwrite(host, 4, (size_t)nv * 3, f);
            if (hipMemcpy(host, d_f, (size_t)nf * 12, hipMemcpyDeviceToHost) != hipSuccess) return 9;
            fwrite(host, 4, (size_t)nf * 3, f);
            if (hipMemcpy(host, d_nrm, (size_t)nv * 12, hipMemcpyDeviceToHost) != hipSuccess) return 9;
            fwrite(host, 4, (size_t)nv * 3, f);
            if (hipMemcpy(host, d_val, (size_t)nv * 4, hipMemcpyDeviceToHost) != hipSuccess) return 9;
            fwrite(host, 4, (size_t)nv, f);
        }
    }
    fclose(f);

    /* ---- the 64-wide networks' WHOLE backward from plain C (ABI v6, nm_mlp_backward_fused): what loss.backward() leaves in the
     *      .grad of every Linear of FlexibleNeRFModel (src/nerf/models.py:60-80) for sum(radiance) over the first 896 = 7 x 128
     *      points as one-sample rays.  The tape holds the view layer's activation rows inside the direction-encoding rows
     *      (d_v = d_enc_dir + 32, v_stride = 64).  Written to <out.bin>.fused: fc_feat.weight, layer1.bias, fc_rgb.weight,
     *      fc_alpha.weight, layers_dir.0.weight */
    {
        const long m = 896;
        char path[4096];
        snprintf(path, sizeof path, "%s.fused", argv[4]);
        FILE* g = fopen(path, "wb");
        if (n >= m && nm_mlp_backward_fused_supported(mlp, m)) {
            const size_t mt = (size_t)m / 16;
            nm_mlp_tape t2 = {0};
            float *d_rad3, *d_ones;
            hipMalloc((void**)&t2.d_h, (size_t)L * m * H * 4); hipMalloc((void**)&t2.d_feat, (size_t)m * H * 4);
            hipMalloc((void**)&t2.d_mask_h, (size_t)L * mt * 64 * 8); hipMalloc((void**)&t2.d_mask_v, mt * 64 * 8);
            hipMalloc((void**)&t2.d_enc_xyz, (size_t)m * 64 * 4); hipMalloc((void**)&t2.d_enc_dir, (size_t)m * 64 * 4);
            t2.d_v = t2.d_enc_dir + 32; t2.v_stride = 64;
            hipMalloc((void**)&d_rad3, (size_t)m * 16); hipMalloc((void**)&d_ones, (size_t)m * 16);
            hipMemcpy(d_ones, ones, (size_t)m * 16, hipMemcpyHostToDevice);
            if (nm_mlp_forward_train(mlp, d_pts, 1, d_dirs, d_t, m, 1, &t2, d_rad3, NULL)) { fprintf(stderr, "forward_train (fused tape): %s\n", nm_last_error()); return 16; }
            nm_mlp_param_grads pg = {0};
            const size_t sizes[6] = {(size_t)H * DX, (size_t)H * H, (size_t)(H / 2) * (H + DD), H, 3 * (H / 2), H};
            hipMalloc((void**)&pg.layer1_weight, sizes[0] * 4); hipMalloc((void**)&pg.layer1_bias, H * 4);
            for (int i = 0; i < L - 1; ++i) { hipMalloc((void**)&pg.xyz_weight[i], sizes[1] * 4); hipMalloc((void**)&pg.xyz_bias[i], H * 4); }
            hipMalloc((void**)&pg.feat_weight, sizes[1] * 4); hipMalloc((void**)&pg.feat_bias, H * 4);
            hipMalloc((void**)&pg.dir_weight, sizes[2] * 4); hipMalloc((void**)&pg.dir_bias, (H / 2) * 4);
            hipMalloc((void**)&pg.alpha_weight, H * 4); hipMalloc((void**)&pg.alpha_bias, 4);
            hipMalloc((void**)&pg.rgb_weight, sizes[4] * 4); hipMalloc((void**)&pg.rgb_bias, 3 * 4);
            void* d_fws;
            hipMalloc(&d_fws, (size_t)nm_mlp_backward_fused_workspace_bytes(mlp));
            if (nm_mlp_backward_fused(mlp, m, &t2, d_rad3, d_ones, NULL, &pg, d_fws, NULL)) { fprintf(stderr, "backward_fused: %s\n", nm_last_error()); return 17; }
            const float* src[5] = {pg.feat_weight, pg.layer1_bias, pg.rgb_weight, pg.alpha_weight, pg.dir_weight};
            const size_t cnt[5] = {sizes[1], H, sizes[4], H, sizes[2]};
            float* host = (float*)malloc(sizes[2] * 4 > sizes[1] * 4 ? sizes[2] * 4 : sizes[1] * 4);
            for (int k = 0; k < 5; ++k) {
                if (hipMemcpy(host, src[k], cnt[k] * 4, hipMemcpyDeviceToHost) != hipSuccess) return 9;
                fwrite(host, 4, cnt[k], g);
            }
        }
        fclose(g);
    }
    printf("abi %d flops/sample %lld ok\n", nm_abi_version(), (long long)nm_mlp_flops_per_sample(mlp, 0));
    nm_mlp_destroy(mlp);
    return 0;
}
