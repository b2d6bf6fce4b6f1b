/* Language-neutral use of the C ABI (include/nerfmeshes_hip.h): no Python, no torch.
 * Builds a 4x64 FlexibleNeRFModel from weights read from a raw fp32 file, evaluates n points with
 * nm_mlp_sample_points and writes the (n,4) result to a raw fp32 file; then runs the training entry points
 * (nm_mlp_forward_train + nm_mlp_backward with dL/d(radiance) = 1) on the same points as one-sample rays and appends
 * the radiance again, delta at layer1's output (n,H) and the head deltas (n,4).  The pytest driver
 * (tests/test_gpu_cabi_c.py) generates the inputs and checks everything against the CPU oracle.
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
    nm_mlp_tape tape;
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
    fclose(f);
    printf("abi %d flops/sample %lld ok\n", nm_abi_version(), (long long)nm_mlp_flops_per_sample(mlp, 0));
    nm_mlp_destroy(mlp);
    return 0;
}
