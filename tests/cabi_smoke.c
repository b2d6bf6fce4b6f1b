/* Language-neutral use of the C ABI (include/nerfmeshes_hip.h): no Python, no torch.
 * Builds a 4x64 FlexibleNeRFModel from weights read from a raw fp32 file, evaluates n points with
 * nm_mlp_sample_points and writes the (n,4) result to a raw fp32 file.  The pytest driver
 * (tests/test_gpu_cabi_c.py) generates the inputs and checks the output against the CPU oracle.
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
    fclose(f);
    printf("abi %d flops/sample %lld ok\n", nm_abi_version(), (long long)nm_mlp_flops_per_sample(mlp, 0));
    nm_mlp_destroy(mlp);
    return 0;
}
