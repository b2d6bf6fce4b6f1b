"""ctypes binding of libnerfmeshes_hip.so (the C ABI declared in include/nerfmeshes_hip.h).

There is deliberately NO fallback: if the shared object is missing or a symbol cannot be bound,
importing the hot path raises.  `load()` only dlopens the library (works without a GPU -- used by
the CPU test that checks every declared symbol is exported); compute calls need a MI355X.
"""
import ctypes as C
import os

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libnerfmeshes_hip.so")

c_float_p = C.POINTER(C.c_float)
c_void_p = C.c_void_p


class MlpDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("num_layers", "hidden_size", "skip_step", "num_encoding_fn_xyz",
                                         "num_encoding_fn_dir", "include_input_xyz", "include_input_dir",
                                         "use_viewdirs")]


class MlpWeights(C.Structure):
    _fields_ = [("layer1_w", c_void_p), ("layer1_b", c_void_p),
                ("layers_xyz_w", C.POINTER(c_void_p)), ("layers_xyz_b", C.POINTER(c_void_p)),
                ("layers_dir0_w", c_void_p), ("layers_dir0_b", c_void_p),
                ("fc_alpha_w", c_void_p), ("fc_alpha_b", c_void_p),
                ("fc_rgb_w", c_void_p), ("fc_rgb_b", c_void_p),
                ("fc_feat_w", c_void_p), ("fc_feat_b", c_void_p),
                ("freq_xyz", c_void_p), ("freq_dir", c_void_p)]


class BundleOut(C.Structure):
    _fields_ = [(n, c_void_p) for n in ("d_rgb_map", "d_depth_map", "d_weights", "d_mask_weights", "d_acc_map",
                                        "d_disp_map")]


class MlpTape(C.Structure):
    _fields_ = [(n, c_void_p) for n in ("d_h", "d_feat", "d_v", "d_mask_h", "d_mask_v", "d_enc_xyz", "d_enc_dir")] + [("v_stride", C.c_int32), ("skip_h0", C.c_int32)]


class MlpDeltas(C.Structure):
    _fields_ = [(n, c_void_p) for n in ("d_h", "d_feat", "d_v", "d_last")]


class MlpParamGrads(C.Structure):          # nm_mlp_param_grads (ABI v6: nm_mlp_backward_fused)
    _fields_ = [("layer1_weight", c_void_p), ("layer1_bias", c_void_p), ("xyz_weight", c_void_p * 8), ("xyz_bias", c_void_p * 8),
                ("feat_weight", c_void_p), ("feat_bias", c_void_p), ("dir_weight", c_void_p), ("dir_bias", c_void_p),
                ("alpha_weight", c_void_p), ("alpha_bias", c_void_p), ("rgb_weight", c_void_p), ("rgb_bias", c_void_p)]


class WeightGradJob(C.Structure):
    _fields_ = [("d_delta", c_void_p), ("d_act", c_void_p), ("d_dw", c_void_p), ("dw_ld", C.c_int32), ("dw_col0", C.c_int32),
                ("d_dbias", c_void_p)]


class BundleGrads(C.Structure):
    _fields_ = [(n, c_void_p) for n in ("d_rgb_map", "d_acc_map", "d_depth_map", "d_weights")]


class RenderCfg(C.Structure):
    _fields_ = [("num_coarse", C.c_int32), ("num_fine", C.c_int32), ("lindisp", C.c_int32),
                ("white_background", C.c_int32), ("training", C.c_int32), ("attenuation_threshold", C.c_float)]


class View(C.Structure):
    _fields_ = [("c2w", C.c_float * 12), ("height", C.c_int32), ("width", C.c_int32), ("focal", C.c_double),
                ("use_ndc", C.c_int32), ("ndc_near", C.c_double)]


# name -> (restype, argtypes); the single source the symbol-export test iterates over.
SIGNATURES = {
    "nm_last_error": (C.c_char_p, []),
    "nm_abi_version": (C.c_int, []),
    "nm_device_count": (C.c_int, []),
    "nm_mlp_create": (C.c_int, [C.POINTER(MlpDesc), C.POINTER(MlpWeights), C.c_int, C.POINTER(c_void_p)]),
    "nm_mlp_create_ex": (C.c_int, [C.POINTER(MlpDesc), C.POINTER(MlpWeights), C.c_int, C.c_int, C.POINTER(c_void_p)]),
    "nm_mlp_precision": (C.c_int, [c_void_p]),
    "nm_mlp_destroy": (None, [c_void_p]),
    "nm_mlp_kernel_variant": (C.c_int, [c_void_p, C.POINTER(C.c_int)]),
    "nm_mlp_flops_per_sample": (C.c_int64, [c_void_p, C.c_int]),
    "nm_mlp_profile_enable": (C.c_int, [C.c_int]),
    "nm_mlp_profile_read": (C.c_int, [C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "nm_mlp_sample_points": (C.c_int, [c_void_p, c_void_p, c_void_p, C.c_int64, c_void_p, c_void_p]),
    "nm_mlp_eval_rays": (C.c_int, [c_void_p, c_void_p, C.c_int, c_void_p, c_void_p, C.c_int64, C.c_int32, c_void_p,
                                   c_void_p]),
    "nm_mlp_grid_query": (C.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                    C.c_int64, C.c_int64, C.c_int32, c_void_p, c_void_p]),
    "nm_ray_bundle": (C.c_int, [c_float_p, C.c_int32, C.c_int32, C.c_float, C.c_int64, C.c_int64, c_void_p,
                                c_float_p, c_void_p]),
    "nm_coarse_intervals": (C.c_int, [c_void_p, c_void_p, c_void_p, C.c_int, C.c_int, C.c_int64, C.c_int32,
                                      c_void_p, c_void_p]),
    "nm_composite": (C.c_int, [c_void_p, c_void_p, c_void_p, C.c_int64, C.c_int32, C.c_float, C.c_int, C.c_int,
                               C.POINTER(BundleOut), c_void_p]),
    "nm_sample_pdf": (C.c_int, [c_void_p, c_void_p, c_void_p, C.c_int64, C.c_int32, C.c_int32, c_void_p, c_void_p]),
    "nm_render_workspace_bytes": (C.c_int64, [C.c_int64, C.c_int32, C.c_int32]),
    "nm_render_rays": (C.c_int, [c_void_p, c_void_p, C.POINTER(RenderCfg), c_void_p, C.c_int, c_void_p, c_void_p,
                                 c_void_p, C.c_int, c_void_p, c_void_p, C.c_int64, c_void_p, C.POINTER(BundleOut),
                                 C.POINTER(BundleOut), c_void_p]),
    "nm_render_view": (C.c_int, [c_void_p, c_void_p, C.POINTER(RenderCfg), C.POINTER(View), C.c_int64, C.c_int64, c_void_p,
                                 c_void_p, C.c_int, c_void_p, c_void_p, c_void_p, C.POINTER(BundleOut),
                                 C.POINTER(BundleOut), c_void_p]),
    "nm_view_rays": (C.c_int, [C.POINTER(View), C.c_int64, C.c_int64, c_void_p, c_void_p, c_void_p]),
    "nm_ndc_rays": (C.c_int, [C.c_int32, C.c_int32, C.c_double, C.c_double, c_void_p, C.c_int, c_void_p, C.c_int64,
                              c_void_p, c_void_p, c_void_p]),
    "nm_positional_encoding": (C.c_int, [c_void_p, C.c_int64, C.c_int32, c_float_p, C.c_int32, C.c_int32, c_void_p,
                                         c_void_p]),
    # training path
    "nm_mlp_refresh": (C.c_int, [c_void_p, C.POINTER(MlpWeights), c_void_p]),
    "nm_mlp_refresh_count": (C.c_int64, [c_void_p]),
    "nm_mlp_tapes_encodings": (C.c_int, [c_void_p]),
    "nm_mlp_weights_current": (C.c_int, [c_void_p, C.POINTER(MlpWeights), c_void_p, C.POINTER(C.c_int32)]),
    "nm_mlp_forward_train": (C.c_int, [c_void_p, c_void_p, C.c_int, c_void_p, c_void_p, C.c_int64, C.c_int32,
                                       C.POINTER(MlpTape), c_void_p, c_void_p]),
    "nm_mlp_backward_fused_supported": (C.c_int, [c_void_p, C.c_int64]),
    "nm_mlp_backward_fused_workspace_bytes": (C.c_int64, [c_void_p]),
    "nm_mlp_backward_fused": (C.c_int, [c_void_p, C.c_int64, C.POINTER(MlpTape), c_void_p, c_void_p, c_void_p, C.POINTER(MlpParamGrads),
                                        c_void_p, c_void_p]),
    "nm_mlp_backward_stops_at_xyz0": (C.c_int, [c_void_p]),
    "nm_mlp_backward_ex": (C.c_int, [c_void_p, C.c_int64, C.POINTER(MlpTape), c_void_p, c_void_p, C.POINTER(MlpDeltas), C.c_int32, c_void_p]),
    "nm_mlp_export_xyz_weight": (C.c_int, [c_void_p, C.c_int32, c_void_p, c_void_p]),
    "nm_mlp_export_layer1_transposed": (C.c_int, [c_void_p, c_void_p, c_void_p]),
    "nm_mlp_linear_layer1_finish": (C.c_int, [c_void_p, c_void_p, C.c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "nm_mlp_backward": (C.c_int, [c_void_p, C.c_int64, C.POINTER(MlpTape), c_void_p, c_void_p, C.POINTER(MlpDeltas),
                                  c_void_p]),
    "nm_encode_samples": (C.c_int, [c_void_p, c_void_p, C.c_int, c_void_p, c_void_p, C.c_int64, C.c_int32, c_void_p,
                                    c_void_p, c_void_p]),
    "nm_encode_samples_strided": (C.c_int, [c_void_p, c_void_p, C.c_int, c_void_p, c_void_p, C.c_int64, C.c_int32, c_void_p,
                                            C.c_int32, c_void_p, C.c_int32, c_void_p]),
    "nm_mlp_num_cus": (C.c_int, [c_void_p]),
    "nm_weight_grad_workspace_bytes": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32]),
    "nm_weight_grad": (C.c_int, [C.c_int, c_void_p, C.c_int32, c_void_p, C.c_int32, C.c_int32, C.c_int64, c_void_p, c_void_p,
                                 C.c_int32, C.c_int32, c_void_p, c_void_p]),
    "nm_weight_grad_workspace_bytes_ex": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "nm_weight_grad_ex": (C.c_int, [C.c_int, c_void_p, C.c_int32, C.c_int32, c_void_p, C.c_int32, C.c_int32, C.c_int64, c_void_p,
                                    c_void_p, C.c_int32, C.c_int32, c_void_p, c_void_p]),
    "nm_weight_grad_batch": (C.c_int, [C.c_int, C.c_int32, C.POINTER(WeightGradJob), C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int64,
                                       c_void_p, c_void_p]),
    "nm_weight_grad_plan": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32)]),
    "nm_head_grad_workspace_bytes_ex": (C.c_int64, [C.c_int32]),
    "nm_head_grad_ex": (C.c_int, [c_void_p, c_void_p, C.c_int32, C.c_int32, C.c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "nm_head_grad_workspace_bytes": (C.c_int64, [C.c_int32]),
    "nm_head_grad": (C.c_int, [c_void_p, c_void_p, C.c_int32, C.c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "nm_perturb_intervals": (C.c_int, [c_void_p, c_void_p, C.c_int64, C.c_int32, c_void_p, c_void_p]),
    "nm_composite_train": (C.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, C.c_int64, C.c_int32, C.c_float, C.c_int,
                                     C.POINTER(BundleOut), c_void_p]),
    "nm_composite_backward": (C.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, C.c_int64, C.c_int32, C.c_int,
                                        C.POINTER(BundleGrads), c_void_p, c_void_p]),
    "nm_sample_pdf_rand": (C.c_int, [c_void_p, c_void_p, c_void_p, C.c_int64, C.c_int32, C.c_int32, c_void_p,
                                     c_void_p]),
    "nm_buff_intersect": (C.c_int, [c_void_p, C.c_int32, c_void_p, C.c_int, c_void_p, C.c_float, C.c_float, c_void_p,
                                    C.c_int64, C.c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "nm_buff_intersect_ex": (C.c_int, [c_void_p, C.c_int32, c_void_p, C.c_int, c_void_p, C.c_float, C.c_float, c_void_p,
                                       C.c_int64, C.c_int32, C.c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "nm_buff_intersect_random": (C.c_int, [c_void_p, C.c_int32, c_void_p, C.c_int, c_void_p, C.c_float, C.c_float, c_void_p,
                                           c_void_p, C.c_int64, C.c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "nm_np_stats_workspace_bytes": (C.c_int64, [C.c_int64]),
    "nm_np_stats": (C.c_int, [c_void_p, C.c_int64, c_void_p, c_float_p, c_void_p]),
    "nm_np_chunk_count": (C.c_int64, [C.c_int64]),
    "nm_np_chunk_sums": (C.c_int, [c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_float, c_void_p,
                                   c_void_p, c_void_p, c_void_p]),
    "nm_np_finish": (C.c_int, [c_void_p, c_void_p, c_void_p, C.c_int64, C.c_int64, C.c_int32, c_void_p, c_float_p, c_void_p]),
    "nm_tree_workspace_bytes": (C.c_int64, [C.c_int32]),
    "nm_tree_integrate": (C.c_int, [c_void_p, c_void_p, c_void_p, C.c_int64, C.c_int32, C.c_int32, c_void_p, c_void_p,
                                    c_void_p]),
    "nm_mc_workspace_bytes": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32]),
    "nm_mc_vertex_scratch_bytes": (C.c_int64, [C.c_int64, C.c_int64]),
    "nm_mc_count": (C.c_int, [c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_double, c_void_p, C.POINTER(C.c_int64),
                              C.POINTER(C.c_int64), c_void_p]),
    "nm_mc_emit": (C.c_int, [c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_double, c_void_p, c_void_p, C.c_int64,
                             C.c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "nm_mc_count_slab": (C.c_int, [c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_int32, C.c_int32, C.c_int32, c_void_p,
                                   C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64), c_void_p]),
    "nm_mc_emit_slab": (C.c_int, [c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_int32, C.c_int32, C.c_int32, c_void_p,
                                  c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, c_void_p, c_void_p, c_void_p,
                                  c_void_p, c_void_p]),
    "nm_export_obj": (C.c_int, [c_void_p, C.c_int64, c_void_p, C.c_int64, c_void_p, C.c_int64, c_void_p, C.c_int64,
                                C.c_char_p]),
}

_lib = None


class HipLibraryError(RuntimeError):
    pass


def load():
    """dlopen the library and bind every declared symbol; raises HipLibraryError loudly."""
    global _lib
    if _lib is not None:
        return _lib
    # torch ships its own libamdhip64; it must be the HIP runtime of the process (device pointers and
    # streams cross the boundary), so make sure it is loaded before our library resolves its DT_NEEDED.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise HipLibraryError(
            f"{LIB_PATH} is missing: build it with `python -m nerfmeshes_amd.build` "
            "(there is no CPU fallback for the nerfmeshes hot path)")
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:
        raise HipLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise HipLibraryError(f"{LIB_PATH} does not export {name}") from e
        fn.restype, fn.argtypes = res, args
    if lib.nm_abi_version() != 6:
        raise HipLibraryError("ABI version mismatch between _lib.py and libnerfmeshes_hip.so")
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().nm_last_error()
        raise HipLibraryError(f"{what} failed (rc={rc}): {msg.decode() if msg else '?'}")
