"""Shared helpers for the parity tests (test infrastructure)."""
import ast
import os

import numpy as np

from nerfmeshes_amd import synthetic as S
from oracle import nerf_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
BUNDLE_KEYS = ("rgb_map", "depth_map", "weights", "mask_weights", "acc_map", "disp_map")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def golden_hparams(g):
    return {str(k): ast.literal_eval(str(v)) for k, v in zip(g["hparams_keys"], g["hparams_vals"])}


def mlp_kwargs(hp, part):
    keys = ("num_layers", "hidden_size", "skip_step", "num_encoding_fn_xyz", "num_encoding_fn_dir")
    return {k: hp[f"models.{part}.{k}"] for k in keys}


def specs_from_hparams(hp):
    sc = O.MLPSpec(**mlp_kwargs(hp, "coarse"))
    sf = O.MLPSpec(**mlp_kwargs(hp, "fine")) if hp["models.use_fine"] else None
    rs = O.RenderSpec(num_coarse=hp["nerf.train.num_coarse"], num_fine=hp["nerf.train.num_fine"],
                      lindisp=hp["nerf.validation.lindisp"], white_background=hp["dataset.white_background"])
    return sc, sf, rs


def golden_weights(g, hp):
    gain, bias = float(g["gain"]), float(g["bias"])
    wc = S.make_mlp_weights(int(g["seed_coarse"]), density_gain=gain, density_bias=bias, **mlp_kwargs(hp, "coarse"))
    wf = None
    if hp["models.use_fine"]:
        wf = S.make_mlp_weights(int(g["seed_fine"]), density_gain=gain, density_bias=bias, **mlp_kwargs(hp, "fine"))
    return wc, wf


RENDER_CASES = ("render_lego_scene", "render_lego_default_init", "render_lego_perray_white_lindisp",
                "render_tiny", "render_fern_8x128")
