/*
 * nerfmeshes_hip.h -- C ABI of the MI355X (gfx950) hot path of qway/nerfmeshes.
 *
 * The reference is pure Python/PyTorch and has no FFI of its own:
 * its boundary is the Python class surface (models.NeRFModel.forward/query/sample_points,
 * nerf.* helpers, mesh_nerf.extract_*).  This header is the C-ABI that sits *under* re-implemented
 * Python classes of the same names (package `nerfmeshes_amd`); every entry point cites the
 * reference code (file:line under /root/reference) whose arithmetic it replaces.
 *
 * Conventions
 *  - every function returns 0 on success, non-zero on failure; nm_last_error() gives the message
 *    (thread-local).  Nothing here ever falls back to a CPU path.
 *  - `d_` pointers are DEVICE pointers (HBM) owned by the caller; `h_` pointers are HOST pointers.
 *    The library allocates nothing on the hot calls; it owns only the packed weights in an
 *    nm_mlp handle.
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream).  Calls are asynchronous.
 *  - all floating point is IEEE fp32 (the reference never casts, train_nerf.py:38-41); index
 *    outputs are int32 (marching cubes faces) or int64 (BuFF voxel ids) as in the reference.
 */
#ifndef NERFMESHES_HIP_H
#define NERFMESHES_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2 (round 3): + nm_mc_count_slab / nm_mc_emit_slab, nm_np_chunk_count / nm_np_chunk_sums / nm_np_finish, use_viewdirs = 0
 * handles; nm_mc_workspace_bytes grew (per-cube decision table, cut-cube list).  Everything in version 1 is unchanged. */
/* 3 (round 4): nm_mlp_create accepts every FlexibleNeRFModel shape (generic kernel family; nm_mlp_kernel_variant >= 1000),
 * NM_KERNEL_GENERIC, training entry points for use_viewdirs = 0 and generic-shape handles (tape / delta fields those do not use
 * may be NULL), nm_mc_emit_slab accepts NULL outputs for an empty own share.  No signature changed; everything in version 2 is
 * unchanged. */
/* 4 (round 5): + nm_weight_grad_ex / nm_head_grad_ex (+ their workspace functions, nm_weight_grad_plan): weight gradients of
 * every layer shape, row stride and sample count; nm_weight_grad / nm_head_grad forward to them (their shape and n % 16
 * restrictions are gone), nm_encode_samples_strided writes whole rows for any stride.  No signature changed; everything in
 * version 3 is unchanged.  nm_mlp_create accepts hidden sizes above 512 and up to 32 encoding functions (layer-wise path). */
/* 5 (round 6): + nm_mlp_refresh_count / nm_mlp_weights_current (the stale-parameter guard), nm_mlp_tape.d_enc_xyz / d_enc_dir
 * (optional: the encoding rows written by the taping forward) + nm_mlp_tapes_encodings.  nm_mlp_tape grew at its end; a
 * zero-initialised struct of the old size keeps its meaning. */
/* 6 (round 6): + nm_mlp_backward_fused (+ _supported, _workspace_bytes, nm_mlp_param_grads): the 64-wide networks' whole
 * backward in one kernel; nm_mlp_tape.v_stride (0 = contiguous rows of d_v, as before); nm_mlp_backward_ex (flags),
 * nm_mlp_export_xyz_weight.  No signature changed. */
#define NM_ABI_VERSION 6

const char* nm_last_error(void);
int nm_abi_version(void);
/* Number of visible HIP devices (<=0: none / error). */
int nm_device_count(void);

/* ------------------------------------------------------------------------------------------
 * FlexibleNeRFModel  (src/nerf/models.py:4-80; PositionalEncoding src/nerf/modules.py:8-37)
 * ------------------------------------------------------------------------------------------ */
typedef struct nm_mlp_desc {
    int32_t num_layers;           /* models.py:7   */
    int32_t hidden_size;          /* models.py:8   */
    int32_t skip_step;            /* models.py:9   */
    int32_t num_encoding_fn_xyz;  /* models.py:10  */
    int32_t num_encoding_fn_dir;  /* models.py:11  */
    int32_t include_input_xyz;    /* models.py:12  */
    int32_t include_input_dir;    /* models.py:13  */
    int32_t use_viewdirs;         /* models.py:16 -- 1: all shipped configs.  0 (models.py:77-79, trunk -> fc_out): fp32, inference
                                   * and training (the tape is the trunk's; see nm_mlp_tape) */
} nm_mlp_desc;

/* Host pointers to the tensors of FlexibleNeRFModel.state_dict(), torch.nn.Linear layout
 * (out_features, in_features) row-major fp32.  layers_xyz_* have num_layers-1 entries.
 * use_viewdirs = 0: the network ends in fc_out (4, hidden_size): pass its colour rows as fc_rgb_w = fc_out.weight (rows
 * 0..2, i.e. 3 x hidden_size contiguous floats) / fc_rgb_b = fc_out.bias, and its density row as fc_alpha_w =
 * fc_out.weight + 3 * hidden_size / fc_alpha_b = fc_out.bias + 3; layers_dir0_*, fc_feat_* and freq_dir are ignored. */
typedef struct nm_mlp_weights {
    const float* layer1_w;            const float* layer1_b;
    const float* const* layers_xyz_w; const float* const* layers_xyz_b;
    const float* layers_dir0_w;       const float* layers_dir0_b;
    const float* fc_alpha_w;          const float* fc_alpha_b;
    const float* fc_rgb_w;            const float* fc_rgb_b;
    const float* fc_feat_w;           const float* fc_feat_b;
    const float* freq_xyz;            /* encode_xyz.frequency_bands (num_encoding_fn_xyz) */
    const float* freq_dir;            /* encode_dir.frequency_bands (num_encoding_fn_dir) */
} nm_mlp_weights;

typedef struct nm_mlp nm_mlp;

/* Packs the weights into the MFMA operand stream and uploads them to `device`.
 * Every shape FlexibleNeRFModel's constructor accepts (models.py:5-58) is served: the shipped configs' shapes (hidden_size
 * 64 / 128 / 256, 6 or 10 xyz and 4 direction functions) by kernels tuned for exactly them, every other one -- any hidden_size
 * up to 512, 0..31 encoding functions per input (32 without the input itself), include_input_* on or off -- by the
 * generic-shape kernel family (padded to the next width class; nm_mlp_kernel_variant reports 1000 + class).  Beyond that -- a
 * hidden_size above 512 or an encoding of more than 48 MFMA k-steps -- the network is evaluated and trained LAYER BY LAYER on the
 * library's general MFMA GEMM (nm_mlp_kernel_variant 2000; the handle then owns a grow-only activation workspace that the
 * first call of a size allocates).  Only more than 32 encoding functions or a weight matrix of more than 2^24 elements fails,
 * with a message.  Every handle trains (generic-shape and layer-wise handles through the tape-row path: masks may be NULL,
 * nm_mlp_tape); NM_PREC_BF16X3 exists for the tuned (shipped) shapes only. */
int nm_mlp_create(const nm_mlp_desc* desc, const nm_mlp_weights* h_weights, int device, nm_mlp** out);

/* Arithmetic of the GEMMs.  NM_PREC_F32 (default, what nm_mlp_create builds): fp32 MFMA, the reference's fp32 arithmetic
 * up to summation order.  NM_PREC_BF16X3 (opt-in; the shipped widths 64 / 128 / 256 with 6 or 10 xyz and 4 direction functions): every product is emulated by six bf16 MFMA
 * products of a three-way split of both operands with fp32 accumulation -- fp32-class error (dropped terms <= 2^-24 of
 * a product) at ~2.3x the throughput, but not bit-comparable with the fp32 path; inference entry points only
 * (nm_mlp_forward_train / nm_mlp_backward refuse such a handle). */
enum { NM_PREC_F32 = 0, NM_PREC_BF16X3 = 1 };
/* OR into `precision` (with NM_PREC_F32): bind the handle to the generic-shape kernel family even where a tuned kernel exists
 * for the shape.  A cross-check, not a mode: the results are the tuned kernel's bit for bit. */
enum { NM_KERNEL_GENERIC = 0x100 };
int nm_mlp_create_ex(const nm_mlp_desc* desc, const nm_mlp_weights* host_weights, int device, int precision, nm_mlp** out);
int nm_mlp_precision(const nm_mlp* mlp);
void nm_mlp_destroy(nm_mlp* mlp);
/* Which tuning variant of the kernel the handle was bound to (0 = production; NM_MLP_VARIANT selects others
 * for A/B measurements) and its waves per workgroup. */
int nm_mlp_kernel_variant(const nm_mlp* mlp, int* waves_per_workgroup);
/* FLOP of one sample through the net (weights-only count). */
int64_t nm_mlp_flops_per_sample(const nm_mlp* mlp, int density_only);

/* Measurement hook (no reference counterpart): while enabled, every launch of the fused MLP kernel is
 * bracketed by hipEvents on its own stream.  nm_mlp_profile_read synchronises those events, returns
 * the number of launches, their summed duration and summed ALGORITHMIC flops, and clears the list. */
int nm_mlp_profile_enable(int on);
int nm_mlp_profile_read(int64_t* launches, double* total_ms, double* total_flops);

/* BaseModel.sample_points / FlexibleNeRFModel.forward  (src/models/model_base.py:65-73,
 * src/nerf/models.py:60-80): d_points (n,3), d_dirs (n,3) -> d_radiance (n,4) = [sigmoid rgb, raw sigma]. */
int nm_mlp_sample_points(nm_mlp* mlp, const float* d_points, const float* d_dirs, int64_t n,
                         float* d_radiance, void* stream);

/* intervals_to_ray_points + model(...) of NeRFModel.forward  (src/models/model_helpers.py:32-35,
 * src/models/model_nerf.py:55-61,67-75): p = o + d * t (two roundings, as torch), view dir = d.
 * d_origins is (1,3) when origins_per_ray == 0, else (rays,3); d_dirs (rays,3); d_t (rays,samples);
 * d_radiance (rays,samples,4). */
int nm_mlp_eval_rays(nm_mlp* mlp, const float* d_origins, int origins_per_ray, const float* d_dirs,
                     const float* d_t, int64_t rays, int32_t samples, float* d_radiance, void* stream);

/* extract_radiance  (src/mesh_nerf.py:37-51): grid point (i,j,k) = (ax0[i], ax1[j], ax2[k]), flattened
 * with k fastest; the point itself is passed as view direction (mesh_nerf.py:45).  Evaluates points
 * [first, first+count) of the flattened grid.  If density_only != 0 writes raw sigma (count,) --
 * the colour branch is skipped -- else writes (count,4). */
int nm_mlp_grid_query(nm_mlp* mlp, const float* d_ax0, const float* d_ax1, const float* d_ax2,
                      int32_t n0, int32_t n1, int32_t n2, int64_t first, int64_t count,
                      int32_t density_only, float* d_out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Ray-batch primitives
 * ------------------------------------------------------------------------------------------ */

/* get_ray_bundle  (src/nerf/nerf_helpers.py:226-277): h_c2w = 12 floats (rows of [:3,:4]);
 * writes unit directions for pixels [first, first+count) of the row-major (H,W) image into
 * d_dirs (count,3), and the origin c2w[:3,3] into h_origin[3]. */
int nm_ray_bundle(const float* h_c2w, int32_t height, int32_t width, float focal, int64_t first,
                  int64_t count, float* d_dirs, float* h_origin, void* stream);

/* A camera view: the inputs of get_ray_bundle (+ DataBundle.ndc, src/data/data_helpers.py:164-165).  Rays of a view
 * can be generated INSIDE the render kernels (nm_render_view below): ray r is pixel first+r of the row-major (H,W)
 * image; no origin / direction buffer is read or written, the host hands over 12 floats. */
typedef struct nm_view {
    float c2w[12];                 /* rows of tform_cam2world[:3,:4] */
    int32_t height, width;
    double focal;                  /* a python float in the reference: enters ndc_rays' constants in fp64 */
    int32_t use_ndc;               /* cfg.dataset.use_ndc: apply ndc_rays(H, W, focal, ndc_near, o, d) to every ray */
    double ndc_near;               /* 1.0 in the reference (data_helpers.py:165) */
} nm_view;

/* Materialise the rays of pixels [first, first+rays): d_origins (rays,3), d_dirs (rays,3) -- get_ray_bundle, then
 * ndc_rays when view->use_ndc (origins become per-ray). */
int nm_view_rays(const nm_view* view, int64_t first_pixel, int64_t rays, float* d_origins, float* d_dirs, void* stream);

/* ndc_rays  (src/nerf/nerf_helpers.py:280-307) on n rays: d_origins (1,3) or (n,3), d_dirs (n,3) ->
 * d_out_origins (n,3), d_out_dirs (n,3); op for op, python-float constants computed in fp64 and rounded once. */
int nm_ndc_rays(int32_t height, int32_t width, double focal, double near_, const float* d_origins, int origins_per_ray,
                const float* d_dirs, int64_t n, float* d_out_origins, float* d_out_dirs, void* stream);

/* PositionalEncoding.forward  (src/nerf/modules.py:26-34) on its own: d_x (n,dim) ->
 * d_out (n, [dim] + 2*dim*num_bands) = [x | sin(x_c * f_k), c-major | cos(...)]; h_bands = frequency_bands (host). */
int nm_positional_encoding(const float* d_x, int64_t n, int32_t dim, const float* h_bands, int32_t num_bands,
                           int32_t include_input, float* d_out, void* stream);

/* RaySampleInterval.forward, deterministic branch  (src/nerf/modules.py:157-186):
 * t[r][k] = near*(1-u[k]) + far*u[k]   (or the lindisp form).  d_u = linspace(0,1,samples) as the
 * module buffer holds it; d_near/d_far are (1,) when bounds_per_ray == 0 else (rays,). */
int nm_coarse_intervals(const float* d_u, const float* d_near, const float* d_far, int bounds_per_ray,
                        int lindisp, int64_t rays, int32_t samples, float* d_t, void* stream);

/* VolumeRenderer.forward + cumprod_exclusive  (src/nerf/modules.py:67-121,
 * src/nerf/nerf_helpers.py:199-223), noise = 0.  Any output pointer may be NULL.
 * d_radiance (rays,samples,4), d_t (rays,samples), d_dirs (rays,3). */
typedef struct nm_bundle_out {
    float* d_rgb_map;      /* (rays,3)       */
    float* d_depth_map;    /* (rays,)        */
    float* d_weights;      /* (rays,samples) */
    float* d_mask_weights; /* (rays,samples) */
    float* d_acc_map;      /* (rays,)        */
    float* d_disp_map;     /* (rays,)        */
} nm_bundle_out;
int nm_composite(const float* d_radiance, const float* d_t, const float* d_dirs, int64_t rays,
                 int32_t samples, float attenuation_threshold, int white_background, int training,
                 const nm_bundle_out* out, void* stream);

/* SamplePDF.forward / sample_pdf, deterministic u  (src/nerf/modules.py:197-248):
 * d_t (rays,coarse), d_weights (rays,coarse), d_u = linspace(0,1,fine) -> d_t_out (rays,coarse+fine) sorted. */
int nm_sample_pdf(const float* d_t, const float* d_weights, const float* d_u, int64_t rays,
                  int32_t coarse, int32_t fine, float* d_t_out, void* stream);

/* NeRFModel.forward  (src/models/model_nerf.py:37-78): the whole coarse -> resample -> fine chain in
 * one call, all intermediates in caller-provided workspace (nm_render_workspace_bytes).  `fine`
 * (and fine_out) may be NULL (models.use_fine False). */
typedef struct nm_render_cfg {
    int32_t num_coarse, num_fine;      /* cfg.nerf.train.num_coarse / num_fine (model_nerf.py:30-31) */
    int32_t lindisp;                   /* cfg.nerf.{train,validation}.lindisp */
    int32_t white_background;          /* cfg.dataset.white_background (model_base.py:31) */
    int32_t training;                  /* module.training: toggles depth_map[acc<1]=0 (modules.py:108) */
    float attenuation_threshold;       /* 1e-5 (model_base.py:32) */
} nm_render_cfg;
int64_t nm_render_workspace_bytes(int64_t rays, int32_t num_coarse, int32_t num_fine);
int nm_render_rays(nm_mlp* coarse, nm_mlp* fine, const nm_render_cfg* cfg, const float* d_origins,
                   int origins_per_ray, const float* d_dirs, const float* d_near, const float* d_far,
                   int bounds_per_ray, const float* d_u_coarse, const float* d_u_fine, int64_t rays,
                   void* d_workspace, const nm_bundle_out* coarse_out, const nm_bundle_out* fine_out,
                   void* stream);

/* The same for rays generated in the kernels from a camera pose (no ray buffers; saves the 24 B/ray read and the
 * host-side `batchify` H2D of src/nerf/nerf_helpers.py:128): pixels [first_pixel, first_pixel+rays) of `view`.
 * Bit-identical to nm_render_rays on the rays nm_view_rays materialises. */
int nm_render_view(nm_mlp* coarse, nm_mlp* fine, const nm_render_cfg* cfg, const nm_view* view, int64_t first_pixel,
                   int64_t rays, const float* d_near, const float* d_far, int bounds_per_ray, const float* d_u_coarse,
                   const float* d_u_fine, void* d_workspace, const nm_bundle_out* coarse_out,
                   const nm_bundle_out* fine_out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Training path: the arithmetic of NeRFModel.training_step
 * (src/models/model_nerf.py:88-151) -- forward with perturb / radiance noise / random resampling
 * (src/nerf/modules.py:82-91, 171-184, 224-228) and what loss.backward() computes for
 * FlexibleNeRFModel.forward (src/nerf/models.py:60-80) and VolumeRenderer.forward
 * (src/nerf/modules.py:67-121).  Random numbers are always the CALLER's device tensors
 * (torch.rand / torch.randn), so a run is reproducible from torch's generator state.
 * ------------------------------------------------------------------------------------------ */

/* Re-pack the network from DEVICE tensors (the live nn.Parameter storages, nn.Linear layout) after an
 * optimizer step: one gather kernel, asynchronous on `stream`, no host round trip.  Pointers as in
 * nm_mlp_weights; freq_xyz / freq_dir are ignored (buffers, fixed at nm_mlp_create). */
int nm_mlp_refresh(nm_mlp* mlp, const nm_mlp_weights* d_weights, void* stream);

/* Stale-parameter guard (ABI v5).  The reference has no packed copy to go stale: its forward reads the nn.Parameter
 * storages themselves (src/nerf/models.py:60-80), so ANY in-place edit -- an optimizer step (src/models/model_base.py:159-162),
 * `p.data.mul_()`, torch._foreach_* on `.data`, load_state_dict -- is visible to the next forward.  A handle holds a packed
 * image instead; these two calls let the host prove that the image still equals the live tensors:
 *   nm_mlp_refresh_count    how many times the image was (re)built since nm_mlp_create (create counts as 1).
 *   nm_mlp_weights_current  *differs = 0 when a checksum of the caller's live tensors (same pointers as nm_mlp_refresh),
 *                           taken through the gather's own index map, equals the checksum the last gather recorded of
 *                           what it packed; 1 otherwise.  One read-only kernel over the parameters (8x256: 2.4 MB) and an
 *                           8-byte read-back: it SYNCHRONISES `stream`.  64-bit sum of bits(value) * (2 * position + 1). */
int64_t nm_mlp_refresh_count(const nm_mlp* mlp);
int nm_mlp_weights_current(nm_mlp* mlp, const nm_mlp_weights* d_weights, void* stream, int32_t* differs);

/* Activations recorded by the training forward, n = rays * samples, tiles = ceil(n / 16), L = num_layers,
 * H = hidden_size.  Rows are the operands of the weight-gradient GEMMs (dW = delta^T @ rows).
 * A use_viewdirs = 0 handle tapes the trunk only: d_feat, d_v and d_mask_v are not touched and may be NULL (likewise the
 * d_feat / d_v of nm_mlp_deltas).  A generic-shape handle (nm_mlp_kernel_variant >= 1000) writes no masks -- its backward
 * kernel reads ReLU' off the activation rows, which nm_mlp_backward therefore needs (d_h, d_feat, d_v) --: d_mask_h and
 * d_mask_v may be NULL. */
typedef struct nm_mlp_tape {
    float* d_h;          /* (L, n, H): [0] layer1 output; [1+i] relu(layers_xyz[i](.))               */
    float* d_feat;       /* (n, H): relu(fc_feat(x))                                               */
    float* d_v;          /* (n, H/2): relu(layers_dir[0](cat(feat, view)))                         */
    uint64_t* d_mask_h;  /* (L, tiles, 64) ReLU masks of layers_xyz[0..L-2] and fc_feat, kernel-private layout */
    uint64_t* d_mask_v;  /* (tiles, 64)                                                            */
    /* ABI v5, optional (NULL: not written).  The PositionalEncoding rows (src/nerf/modules.py:26-34) of every sample point /
     * view direction, 64 floats per sample in the reference's column order [x | sin c-major | cos c-major] -- what the layer1 /
     * skip / view weight gradients contract with.  The taping kernel has these values in registers anyway; writing them here
     * saves the separate nm_encode_samples_strided pass of the backward.  Tuned-family handles only (nm_mlp_kernel_variant
     * < 1000, both encodings <= 64 wide); other handles leave the buffers untouched -- ask nm_mlp_tapes_encodings().
     * Columns beyond the encoding's width are not written. */
    float* d_enc_xyz;    /* (n, 64) or NULL                                                         */
    float* d_enc_dir;    /* (n, 64) or NULL                                                         */
    /* ABI v6: floats between consecutive rows of d_v; 0 = H/2 (contiguous rows).  Tuned-family handles only.  The 64-wide
     * networks' fused backward wants the 32-float rows of d_v INSIDE the direction-encoding rows -- d_v = d_enc_dir + 32,
     * v_stride = 64: an encoding row is 27 floats of its 64 -- so that one block of rows feeds both layers_dir[0]'s and fc_rgb's
     * weight gradients (nm_mlp_backward_fused). */
    int32_t v_stride;
    /* ABI v6: non-zero = do not write d_h[0] (layer1's output; the plane stays allocated, untouched).  For a backward that takes
     * layer1's and layers_xyz[0]'s gradients by linearity (nm_mlp_backward_ex + NM_BACKWARD_STOP_AT_XYZ0, nm_mlp_backward_fused):
     * neither reads it -- a tenth of the tape's bytes.  Tuned-family handles only. */
    int32_t skip_h0;
} nm_mlp_tape;
/* 1 when nm_mlp_forward_train on this handle fills d_enc_xyz / d_enc_dir (when given), 0 when the caller still needs
 * nm_encode_samples_strided. */
int nm_mlp_tapes_encodings(const nm_mlp* mlp);

/* dL/d(pre-activation) of every layer, written by nm_mlp_backward (same row layout as the tape). */
typedef struct nm_mlp_deltas {
    float* d_h;          /* (L, n, H): [0] at layer1's output; [1+i] at layers_xyz[i]'s pre-activation */
    float* d_feat;       /* (n, H) at fc_feat's pre-activation                                       */
    float* d_v;          /* (n, H/2) at layers_dir[0]'s pre-activation                               */
    float* d_last;       /* (n, 4): at fc_rgb's output (pre-sigmoid) x3, at fc_alpha's output        */
} nm_mlp_deltas;

/* FlexibleNeRFModel.forward over ray samples (as nm_mlp_eval_rays) that also records the tape. */
int nm_mlp_forward_train(nm_mlp* mlp, const float* d_origins, int origins_per_ray, const float* d_dirs,
                         const float* d_t, int64_t rays, int32_t samples, const nm_mlp_tape* tape,
                         float* d_radiance, void* stream);

/* Back-propagation through the network: d_grad_radiance (n,4) = dL/d(sigmoid(rgb), sigma) and the
 * forward's own d_radiance (n,4) -> deltas.  The weight gradients are products of those rows with the tape rows:
 * grad(layers_xyz[i].weight) = deltas.d_h[1+i]^T @ tape.d_h[i] (@ the encoding rows for the skip columns), grad(bias) =
 * column sums -- nm_weight_grad_ex / nm_head_grad_ex below compute them for every shape (nerfmeshes_amd/train_ops.py:
 * backward lists all of them; tests/cabi_smoke.c does one from plain C). */
int nm_mlp_backward(nm_mlp* mlp, int64_t n, const nm_mlp_tape* tape, const float* d_radiance,
                    const float* d_grad_radiance, const nm_mlp_deltas* deltas, void* stream);

/* ABI v6.  nm_mlp_backward with flags.  NM_BACKWARD_STOP_AT_XYZ0: the chain ends at layers_xyz[0]'s pre-activation -- d_h[1] is
 * the last delta written, d_h[0] is NOT produced, one of the L + 1 transposed layers is never applied (an eighth of the kernel at
 * 8 layers).  layer1 has no activation (src/nerf/models.py:62), so the delta at its output is linear in d_h[1] and its gradients
 * follow from sums over the samples:  grad(layer1.weight) = W0^T (d_h[1]^T @ encoding rows),  grad(layer1.bias) = W0^T (column
 * sums of d_h[1]),  W0 = layers_xyz[0].weight's hidden columns -- two calls of nm_weight_grad_ex, the second over H rows
 * (nerfmeshes_amd/train_ops.py: backward).  Tuned-family handles with num_layers >= 3 (nm_mlp_backward_stops_at_xyz0).
 * nm_mlp_export_xyz_weight writes layers_xyz[layer].weight, (H, H) or (H, H + dx) row-major, out of the handle's PACKED image --
 * the values the kernels of this forward / backward pair used, whatever happened to the live tensors since. */
#define NM_BACKWARD_STOP_AT_XYZ0 1
int nm_mlp_backward_stops_at_xyz0(const nm_mlp* mlp);
int nm_mlp_backward_ex(nm_mlp* mlp, int64_t n, const nm_mlp_tape* tape, const float* d_radiance,
                       const float* d_grad_radiance, const nm_mlp_deltas* deltas, int32_t flags, void* stream);
int nm_mlp_export_xyz_weight(nm_mlp* mlp, int32_t layer, float* d_out, void* stream);
/* The same identity gives layers_xyz[0]'s own weight gradient without its activation rows: tape.d_h[0] = layer1(enc) = W1 enc + b1, so
 *     grad(layers_xyz[0].weight) = d_h[1]^T @ tape.d_h[0] = [d_h[1]^T enc | sum d_h[1]] @ [W1 | b1]^T
 * -- the (H, dx + 1) matrix of sums the caller already has, times (dx + 1, H) = nm_mlp_export_layer1_transposed: rows 0 .. dx - 1
 * hold layer1.weight^T, row dx layer1.bias, again out of the packed image.  One of the L hidden x hidden weight-gradient products
 * over all samples becomes a product over dx + 1 rows. */
int nm_mlp_export_layer1_transposed(nm_mlp* mlp, float* d_out, void* stream);
/* Both products in one launch, the parameters read out of the packed image: d_sums (H, dx) with row stride ld = d_h[1]^T @ enc,
 * d_colsum (H) = the column sums of d_h[1]  ->  d_l1w (H, dx), d_l1b (H) = grad(layer1), d_x0w (H, H) = grad(layers_xyz[0].weight).
 * What nerfmeshes_amd/train_ops.py: backward calls behind the weight-gradient batch. */
int nm_mlp_linear_layer1_finish(nm_mlp* mlp, const float* d_sums, int32_t ld, const float* d_colsum, float* d_l1w, float* d_l1b,
                                float* d_x0w, void* stream);

/* ABI v6.  The whole back-propagation of a 64-wide network in ONE kernel (nerf_bwd_fused.hip): the delta chain of
 * nm_mlp_backward AND the weight / bias gradients of layer1, layers_xyz[*], fc_feat and layers_dir[0] -- what loss.backward()
 * leaves in the .grad of those parameters under NeRFModel.training_step (src/models/model_nerf.py:88-151, through
 * FlexibleNeRFModel.forward, src/nerf/models.py:60-80) -- without a delta row ever reaching HBM (the tape is read once, the
 * per-workgroup partials of all products are added up by one order-fixed reduction: deterministic).  The two 4-row heads
 * (fc_alpha, fc_rgb) are products of the same kernel: their delta d_last (n, 4) is staged in LDS next to the layers' deltas; it
 * is also written to d_last when that is not NULL (as nm_mlp_backward does).  The tape must hold the view layer's activation
 * rows inside the direction-encoding rows (nm_mlp_tape.v_stride: d_v = d_enc_dir + 32, v_stride = 64).
 * Served: tuned-family fp32 handles with hidden_size 64, use_viewdirs = 1, 2 <= num_layers <= 8, at most one skip layer, that
 * tape their encodings (nm_mlp_tapes_encodings; tape->d_enc_xyz / d_enc_dir must be given), n a multiple of 128 -- ask
 * nm_mlp_backward_fused_supported; everything else takes nm_mlp_backward + nm_weight_grad_batch.  Wider networks do not fit:
 * the accumulators of all products must stay resident per workgroup (128 wide: 630 KB, a CU's register file is 512 KB).
 * Gradient buffers are written in the parameters' own shapes: layer1 (H, dx), layers_xyz[i] (H, H) or (H, H + dx) for the
 * skip layer, fc_feat (H, H), layers_dir[0] (H / 2, H + dd), biases (rows,). */
#define NM_FUSED_MAX_LAYERS 8
typedef struct nm_mlp_param_grads {
    float* layer1_weight;
    float* layer1_bias;
    float* xyz_weight[NM_FUSED_MAX_LAYERS];   /* layers_xyz[0 .. num_layers - 2] */
    float* xyz_bias[NM_FUSED_MAX_LAYERS];
    float* feat_weight;
    float* feat_bias;
    float* dir_weight;
    float* dir_bias;
    float* alpha_weight;                      /* fc_alpha (1, H), (1,) */
    float* alpha_bias;
    float* rgb_weight;                        /* fc_rgb (3, H / 2), (3,) */
    float* rgb_bias;
} nm_mlp_param_grads;
int nm_mlp_backward_fused_supported(const nm_mlp* mlp, int64_t n);
int64_t nm_mlp_backward_fused_workspace_bytes(const nm_mlp* mlp);
int nm_mlp_backward_fused(nm_mlp* mlp, int64_t n, const nm_mlp_tape* tape, const float* d_radiance,
                          const float* d_grad_radiance, float* d_last, const nm_mlp_param_grads* grads, void* d_workspace,
                          void* stream);

/* PositionalEncoding.forward (src/nerf/modules.py:26-34) of the sample points / view directions as
 * rows in the reference's column order: d_enc_xyz (n, 3+6*Fx), d_enc_dir (n, 3+6*Fd); either may be NULL. */
int nm_encode_samples(nm_mlp* mlp, const float* d_origins, int origins_per_ray, const float* d_dirs,
                      const float* d_t, int64_t rays, int32_t samples, float* d_enc_xyz, float* d_enc_dir,
                      void* stream);

/* The same with explicit row strides (floats per row >= encoding width): whole rows are written, zero padding up to the
 * stride included (64-float rows are what the tuned nm_weight_grad kernel streams; any stride serves nm_weight_grad_ex, a
 * multiple of 4 floats gives it 16-byte DMA pieces).  d_enc_dir must be NULL for use_viewdirs = 0 handles. */
int nm_encode_samples_strided(nm_mlp* mlp, const float* d_origins, int origins_per_ray, const float* d_dirs,
                              const float* d_t, int64_t rays, int32_t samples, float* d_enc_xyz, int32_t stride_xyz,
                              float* d_enc_dir, int32_t stride_dir, void* stream);

/* Weight and bias gradient of one Linear from tape rows -- what autograd's addmm backward computes for every layer of
 * FlexibleNeRFModel (src/nerf/models.py:60-80):  d_dw[o * dw_ld + dw_col0 + c] = sum_n delta[n][o] * act[n][c] for
 * c < in_features, and (d_dbias != NULL) d_dbias[o] = sum_n delta[n][o].  d_delta (n, out_features) and d_act
 * (n, act_stride) are row-major; n must be a multiple of 16, out_features and act_stride multiples of 64 (activation rows
 * zero-padded beyond in_features).  Supported (out_features, act_stride): (256,256) (256,64) (128,256) (128,128)
 * (128,64) (64,128).  fp32 MFMA, split over the samples across one workgroup per CU, order-fixed reduction of the
 * per-workgroup partials (deterministic; no atomics).  Workspace: nm_weight_grad_workspace_bytes. */
int nm_mlp_num_cus(const nm_mlp* mlp);
int64_t nm_weight_grad_workspace_bytes(int32_t out_features, int32_t act_stride, int32_t num_cus);
int nm_weight_grad(int num_cus, const float* d_delta, int32_t out_features, const float* d_act, int32_t act_stride,
                   int32_t in_features, int64_t n, void* d_workspace, float* d_dw, int32_t dw_ld, int32_t dw_col0,
                   float* d_dbias, void* stream);

/* The 4-row head gradients (fc_alpha: row 3 of dlast^T h[L-1]; fc_rgb: rows 0..2 of dlast^T v -- models.py:74-78's two
 * Linear layers share the delta dlast (n, 4)): d_dw[r * in_features + k] = sum_n dlast[n][r] * act[n][k], d_dbias[r] =
 * sum_n dlast[n][r] (may be NULL).  in_features 64, 128 or 256.  HBM-bound VALU kernel + the order-fixed partial reduction. */
int64_t nm_head_grad_workspace_bytes(int32_t in_features);
int nm_head_grad(const float* d_dlast, const float* d_act, int32_t in_features, int64_t n, void* d_workspace,
                 float* d_dw, float* d_dbias, void* stream);

/* The general forms (ABI v4): ANY out_features / in_features (what FlexibleNeRFModel's constructor accepts,
 * src/nerf/models.py:5-58), ANY row strides (floats; >= the feature counts), ANY n >= 1 -- no padding of rows or columns is
 * required or written: columns beyond a width only reach dW entries nobody reads, rows beyond n are read as zeros through
 * the buffer descriptor's exact extent.  Same dataflow as nm_weight_grad (fp32 MFMA, operands DMA'd HBM -> LDS as they lie
 * in memory, one workgroup per CU, order-fixed reduction: deterministic), its geometry -- tiles per wave, how the 8 waves of
 * a workgroup split the output block and the samples, how many blocks a wide output is cut into -- picked from the shape
 * alone by a planner (nm_weight_grad_plan reports it: [nba, nbb, wa, wb, wk, ta, tb, rows per chunk]).  nm_weight_grad and
 * nm_head_grad forward here for every shape / sample count their tuned kernels do not serve, so no weight gradient of any
 * network the library accepts is a library GEMM. */
int64_t nm_weight_grad_workspace_bytes_ex(int32_t out_features, int32_t delta_stride, int32_t in_features, int32_t act_stride,
                                          int32_t num_cus);
int nm_weight_grad_ex(int num_cus, const float* d_delta, int32_t out_features, int32_t delta_stride, const float* d_act,
                      int32_t in_features, int32_t act_stride, int64_t n, void* d_workspace, float* d_dw, int32_t dw_ld,
                      int32_t dw_col0, float* d_dbias, void* stream);
int nm_weight_grad_plan(int32_t out_features, int32_t delta_stride, int32_t in_features, int32_t act_stride,
                        int32_t aligned16, int32_t num_cus, int32_t* plan8);

/* Several products of ONE shape, stride pair and row count in one launch + one reduction -- the same-shape layers of a network
 * (models.py:63-70: layers_xyz[*] and fc_feat are all hidden x hidden).  A job gets 1 / jobs of the CUs and jobs times the
 * samples per workgroup: the same matrix work, but jobs times fewer per-workgroup partials to write and to reduce (eight
 * 256 x 256 layers: 64 MB instead of 512 MB) and 2 launches instead of 2 * jobs.  At most 16 jobs; the workspace of ONE
 * product (nm_weight_grad_workspace_bytes_ex) suffices: it holds max(1, num_cus / output blocks) sample parts, and a batch
 * that needs more (layers wider than ~1000: the output is cut into many blocks) is issued in sub-batches that re-use it, one
 * after the other on the stream.  Deterministic; the summation grouping differs from jobs separate
 * calls (fewer, longer sample parts). */
typedef struct nm_weight_grad_job {
    const float* d_delta;    /* (n, out_features), row stride delta_stride */
    const float* d_act;      /* (n, in_features), row stride act_stride    */
    float* d_dw;             /* d_dw[o * dw_ld + dw_col0 + c]               */
    int32_t dw_ld, dw_col0;
    float* d_dbias;          /* (out_features,) or NULL                     */
} nm_weight_grad_job;
int nm_weight_grad_batch(int num_cus, int32_t jobs, const nm_weight_grad_job* job, int32_t out_features, int32_t delta_stride,
                         int32_t in_features, int32_t act_stride, int64_t n, void* d_workspace, void* stream);
int64_t nm_head_grad_workspace_bytes_ex(int32_t in_features);
int nm_head_grad_ex(const float* d_dlast, const float* d_act, int32_t in_features, int32_t act_stride, int64_t n,
                    void* d_workspace, float* d_dw, float* d_dbias, void* stream);

/* RaySampleInterval.forward's stratified jitter (src/nerf/modules.py:171-184): d_rand (rays,samples) in [0,1). */
int nm_perturb_intervals(const float* d_t, const float* d_rand, int64_t rays, int32_t samples, float* d_t_out,
                         void* stream);

/* VolumeRenderer.forward in training mode: sigma + d_noise (rays,samples; NULL = no noise) before the
 * ReLU (src/nerf/modules.py:82-93), depth_map not zeroed (modules.py:108). */
int nm_composite_train(const float* d_radiance, const float* d_t, const float* d_dirs, const float* d_noise,
                       int64_t rays, int32_t samples, float attenuation_threshold, int white_background,
                       const nm_bundle_out* out, void* stream);

/* Upstream gradients of the bundle (any may be NULL = zero); disp_map / mask_weights are not differentiable here. */
typedef struct nm_bundle_grads {
    const float* d_rgb_map;   /* (rays,3)       */
    const float* d_acc_map;   /* (rays,)        */
    const float* d_depth_map; /* (rays,)        */
    const float* d_weights;   /* (rays,samples) */
} nm_bundle_grads;
int nm_composite_backward(const float* d_radiance, const float* d_t, const float* d_dirs, const float* d_noise,
                          int64_t rays, int32_t samples, int white_background, const nm_bundle_grads* grads,
                          float* d_grad_radiance, void* stream);

/* SamplePDF.forward with per-ray random u (src/nerf/modules.py:224-228): d_u (rays,fine). */
int nm_sample_pdf_rand(const float* d_t, const float* d_weights, const float* d_u, int64_t rays, int32_t coarse,
                       int32_t fine, float* d_t_out, void* stream);

/* numpy's fp32 statistics of a device array, bit for bit (src/mesh_nerf.py:56-65 picks the marching-cubes level from
 * density.min() / .max() / .std() / .mean() of a numpy fp32 array; the mesh topology depends on that level):
 * h_out6 = [sum, mean, var, std, min, max] as numpy computes them (8192-element buffer chunks accumulated sequentially,
 * pairwise summation with 8 interleaved accumulators inside a chunk, fp32 throughout).  Synchronises the stream. */
int64_t nm_np_stats_workspace_bytes(int64_t n);
int nm_np_stats(const float* d_x, int64_t n, void* d_workspace, float* h_out6, void* stream);

/* The same statistics when the array is spread over several devices (multi-GPU mesh extraction: every rank holds an axis-0
 * slab of the density grid and the adaptive iso level of src/mesh_nerf.py:56-65 still has to be numpy's, bit for bit).
 * numpy sums 8192-element chunks pairwise and accumulates the chunk sums sequentially, so the two stages separate:
 *   nm_np_chunk_sums  a rank holding the global elements [first, first + count) sums the chunks [chunk_lo, chunk_hi) of
 *                     the GLOBAL array of n elements (all of their elements must be held).  First pass
 *                     (squared_deviation = 0): sums + per-chunk min / max (NaN-propagating, as numpy); second pass
 *                     (squared_deviation = 1, with the mean of the first): sums of (x - mean)^2; d_csum then needs room
 *                     for chunk_hi - chunk_lo + 1 floats.
 *   nm_np_finish      the chunk sums of all ranks concatenated in chunk order -> [sum, mean, -, -, min, max] (first pass)
 *                     or [-, -, var, std, -, -] (second pass) in d_out6 / h_out6; synchronises the stream. */
int64_t nm_np_chunk_count(int64_t n);
int nm_np_chunk_sums(const float* d_x, int64_t first, int64_t count, int64_t n, int64_t chunk_lo, int64_t chunk_hi,
                     int32_t squared_deviation, float mean, float* d_csum, float* d_cmin, float* d_cmax, void* stream);
int nm_np_finish(const float* d_csum, const float* d_cmin, const float* d_cmax, int64_t chunks, int64_t n,
                 int32_t squared_deviation, float* d_out6, float* h_out6, void* stream);

/* ------------------------------------------------------------------------------------------
 * BuFF voxel-tree sampler: TreeSampling.batch_ray_voxel_intersect, deterministic branch
 * (src/nerf/tree.py:215-343).  d_voxels (nvox,2,3) min/max corners; d_origins (1,3) or (rays,3);
 * d_u = linspace(0,1,samples).  Outputs: d_z (rays,samples) sorted depths, d_idx (rays,samples) int64
 * voxel ids, d_mask (rays,) uint8 "ray crosses at least one voxel inside [near, far]".
 * Synchronises the stream (overflow check: a ray may cross at most 512 voxels).
 * ------------------------------------------------------------------------------------------ */
int nm_buff_intersect(const float* d_voxels, int32_t nvox, const float* d_origins, int origins_per_ray,
                      const float* d_dirs, float near_, float far_, const float* d_u, int64_t rays,
                      int32_t samples, float* d_z, int64_t* d_idx, uint8_t* d_mask, void* stream);

/* The same with an explicit tie order for the three sorts of the reference (src/nerf/tree.py:300,306,335 call
 * torch.sort with its UNSTABLE default):
 *   NM_TIES_STABLE     ties keep voxel-index / sample order: every id is the voxel its sample lies in (default);
 *   NM_TIES_REFERENCE  ties ordered as torch's CPU sort orders them (libstdc++ std::sort over (key, index) pairs,
 *                      restated in the kernel): d_idx equals the reference's output bit for bit, including its
 *                      scrambled attribution of samples to the crossed voxels.  The three introsorts are evaluated
 *                      wave-parallel and exactly (one wavefront per ray; 7.6 ms vs 1.3 ms per 65 536 rays x 192
 *                      samples on 1728 voxels); nvox <= 8192, at most 512 crossed voxels per ray (error 4 beyond).
 *                      Synchronises the stream.  d_z and d_mask are identical in both. */
enum { NM_TIES_STABLE = 0, NM_TIES_REFERENCE = 1 };
int nm_buff_intersect_ex(const float* d_voxels, int32_t nvox, const float* d_origins, int origins_per_ray,
                         const float* d_dirs, float near_, float far_, const float* d_u, int64_t rays,
                         int32_t samples, int32_t tie_order, float* d_z, int64_t* d_idx, uint8_t* d_mask, void* stream);

/* The `tree.use_random_sampling` branch of the same function (src/nerf/tree.py:280-297, :337-341) as a function of the
 * draws: d_u_pick (rays,samples) DOUBLE = the uniforms torch.multinomial(weights, samples, replacement=True) consumes
 * (weights 1 for crossed voxels, 1e-12 otherwise: an inverse-CDF pick among the crossed voxels in index order, decided
 * in fp32 as ATen does), d_u_pos (rays,samples) float = torch.rand_like(values_min) (the position inside the drawn
 * voxel's [t_enter, t_exit]).  Outputs as above; given the reference's own draws they equal its output bit for bit on
 * every ray that crosses a voxel (one exception: a draw u below the 1e-12 weights' own share, u < 1.7e-9, selects a
 * NON-crossed voxel in the reference and the first crossed one here).  Rows of rays that cross nothing are zero-filled (d_mask 0): the reference samples
 * arbitrary voxels there, its caller overwrites those depths (src/models/model_buff.py:53) and nothing reads the ids.
 * Synchronises the stream (overflow check). */
int nm_buff_intersect_random(const float* d_voxels, int32_t nvox, const float* d_origins, int origins_per_ray,
                             const float* d_dirs, float near_, float far_, const double* d_u_pick,
                             const float* d_u_pos, int64_t rays, int32_t samples, float* d_z, int64_t* d_idx,
                             uint8_t* d_mask, void* stream);

/* TreeSampling.ray_batch_integration (src/nerf/tree.py:177-206), training-time tree maintenance: the samples of
 * the rays that hit the tree -- d_idx / d_weights / d_mask_weights, `count` elements each (= indices[mask],
 * weights[mask], mask_weights[mask] flattened) -- update the running voxel weights d_memm (nvox,) in place:
 * memm[v] += (sum_w[v] / sum_mask[v] - memm[v]) / counter wherever sum_mask[v] > 0.  `counter` is the tree's
 * 1-based integration counter (the caller increments it).  Workspace: nm_tree_workspace_bytes(nvox). */
int64_t nm_tree_workspace_bytes(int32_t nvox);
int nm_tree_integrate(const int64_t* d_idx, const float* d_weights, const float* d_mask_weights, int64_t count,
                      int32_t nvox, int32_t counter, float* d_memm, void* d_workspace, void* stream);


/* ------------------------------------------------------------------------------------------
 * Marching cubes: skimage.measure.marching_cubes(volume, level) as called at src/mesh_nerf.py:79
 * (third-party scikit-image 0.17.2, Lewiner MC33, defaults: step_size=1, gradient_direction='descent',
 * allow_degenerate=True, no mask).  Output-identical: vertices (V,3) fp32 in (axis0,axis1,axis2) order,
 * faces (F,3) int32 (same vertex numbering and triangle order), normals (V,3), values (V,).
 * Two-phase because V and F are data dependent:
 *   nm_mc_count  classifies every cube and returns V and F (synchronises the stream);
 *   nm_mc_emit   writes the four arrays (caller-allocated from those counts).
 * d_workspace (nm_mc_workspace_bytes) must be kept between the two calls; d_vertex_scratch
 * (nm_mc_vertex_scratch_bytes(V, F)) is only used by nm_mc_emit.  V == 0 is skimage's
 * RuntimeError('No surface found at the given iso value.'); the level-in-range ValueError is the
 * host wrapper's check, as in skimage's Python wrapper.
 * ------------------------------------------------------------------------------------------ */
int64_t nm_mc_workspace_bytes(int32_t n0, int32_t n1, int32_t n2);
int64_t nm_mc_vertex_scratch_bytes(int64_t vertices, int64_t faces);
int nm_mc_count(const float* d_volume, int32_t n0, int32_t n1, int32_t n2, double iso, void* d_workspace,
                int64_t* h_vertices, int64_t* h_faces, void* stream);
int nm_mc_emit(const float* d_volume, int32_t n0, int32_t n1, int32_t n2, double iso, void* d_workspace,
               void* d_vertex_scratch, int64_t vertices, int64_t faces, float* d_verts, int32_t* d_faces,
               float* d_normals, float* d_values, void* stream);

/* Per-slab marching cubes (multi-GPU mesh extraction: every rank meshes its own axis-0 slab, only the triangles are
 * gathered -- SURVEY.md 8(e), BASELINE.json north_star "all-gather of emitted triangles"; replaces the all-gather of the
 * density grid in front of skimage.measure.marching_cubes, src/mesh_nerf.py:79).
 * d_volume holds the global planes [z_global, z_global + n0) of the grid.  Its first cube layer is a ghost of the slab below
 * when ghost_below = 1 (classified and numbered -- the faces of the layer above reference the vertices it creates -- but
 * nothing of it is emitted), its last one a ghost of the slab above when ghost_above = 1 (never classified; its voxels are
 * read when the normals of this slab's top vertices replay the cubes around them).  Vertex ownership follows the GLOBAL
 * plane index, so a slab creates exactly the vertices the single-volume run creates in its layers, in the same order.
 *   nm_mc_count_slab  -> V, F including the ghost layer's, and the ghost layer's own V_g, F_g;
 *   nm_mc_emit_slab   writes the slab's V - V_g vertices / normals / values and F - F_g faces; face entries are
 *                     local id + index_base: with index_base = (vertices of all lower slabs) - V_g the concatenation of
 *                     the ranks' arrays in rank order IS the single-volume mesh, bit for bit, vertex numbering included.
 * With z_global = ghost_below = ghost_above = 0 these are nm_mc_count / nm_mc_emit. */
int nm_mc_count_slab(const float* d_volume, int32_t n0, int32_t n1, int32_t n2, double iso, int32_t z_global,
                     int32_t ghost_below, int32_t ghost_above, void* d_workspace, int64_t* h_vertices, int64_t* h_faces,
                     int64_t* h_ghost_vertices, int64_t* h_ghost_faces, void* stream);
int nm_mc_emit_slab(const float* d_volume, int32_t n0, int32_t n1, int32_t n2, double iso, int32_t z_global,
                    int32_t ghost_below, int32_t ghost_above, void* d_workspace, void* d_vertex_scratch, int64_t vertices,
                    int64_t faces, int64_t ghost_vertices, int64_t ghost_faces, int64_t index_base, float* d_verts,
                    int32_t* d_faces, float* d_normals, float* d_values, void* stream);

/* ------------------------------------------------------------------------------------------
 * Mesh export: export_obj (src/nerf/nerf_helpers.py:86-111), byte-identical text.  HOST arrays:
 * vertices (V,3), diffuse (C,3) with C <= V allowed (colours only while they last), normals (N,3),
 * triangles (F,3) int32, 0-based.  Numbers are printed as Python prints repr(float(x)).  Multi-threaded.
 * ------------------------------------------------------------------------------------------ */
int nm_export_obj(const float* h_vertices, int64_t num_vertices, const float* h_diffuse, int64_t num_diffuse,
                  const float* h_normals, int64_t num_normals, const int32_t* h_triangles, int64_t num_triangles,
                  const char* path);

#ifdef __cplusplus
}
#endif
#endif /* NERFMESHES_HIP_H */
