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


def golden_part(g, prefix):
    """The entries `prefix.*` of a fixture that holds several cases, with the prefix removed."""
    return {k[len(prefix) + 1:]: g[k] for k in g.files if k.startswith(prefix + ".")}


def mlp_kwargs(hp, part):
    keys = ("num_layers", "hidden_size", "skip_step", "num_encoding_fn_xyz", "num_encoding_fn_dir")
    return {k: hp[f"models.{part}.{k}"] for k in keys}


def specs_from_hparams(hp):
    sc = O.MLPSpec(**mlp_kwargs(hp, "coarse"))
    sf = O.MLPSpec(**mlp_kwargs(hp, "fine")) if hp["models.use_fine"] else None
    rs = O.RenderSpec(num_coarse=hp["nerf.train.num_coarse"], num_fine=hp["nerf.train.num_fine"],
                      lindisp=hp["nerf.validation.lindisp"], white_background=hp["dataset.white_background"])
    return sc, sf, rs


def gen_weights(seed, gain, bias, **kw):
    """Mirror of tests/golden/make_golden.py::gen_weights (gain == 0 -> calibrated smooth scene)."""
    if float(gain) == 0:
        return S.make_scene_weights(int(seed), **kw)
    return S.make_mlp_weights(int(seed), density_gain=float(gain), density_bias=float(bias), **kw)


def golden_weights(g, hp):
    wc = gen_weights(g["seed_coarse"], g["gain"], g["bias"], **mlp_kwargs(hp, "coarse"))
    wf = None
    if hp["models.use_fine"]:
        wf = gen_weights(g["seed_fine"], g["gain"], g["bias"], **mlp_kwargs(hp, "fine"))
    return wc, wf


RENDER_CASES = ("render_lego_scene", "render_lego_rough", "render_lego_default_init", "render_lego_perray_white_lindisp",
                "render_tiny", "render_fern_8x128")


def well_conditioned_rays(g):
    """Rays whose hierarchical resampling is numerically meaningful.  SamplePDF normalises
    (weights[1:-1] + 1e-5); when the coarse pass saw (almost) nothing -- sum of coarse weights within ~10x of
    that 62 * 1e-5 floor -- the pdf is dominated by the fp32 noise (~1e-4 absolute) of individual coarse weights,
    every fine sample moves, and a grazing sliver of density is integrated differently.  The reference shows
    the same sensitivity to its own summation order; such rays are compared loosely."""
    if "fine.rgb_map" not in g.files:
        return np.ones(g["coarse.acc_map"].shape, dtype=bool)
    acc = g["coarse.acc_map"]
    return ~((acc > 0) & (acc < 5e-3))    # exactly-empty rays (all coarse weights 0) are perfectly stable
