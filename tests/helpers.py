"""Shared helpers for the parity tests (test infrastructure)."""
import ast
import os
import zlib

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


def grad_digest(name, grad):
    """Compact fingerprint of one gradient tensor (a 1.19 M-parameter gradient set would be a 4.8 MB fixture): fp64 sum,
    L2 norm, largest magnitude, the projection onto a standard-normal vector seeded by the tensor's NAME (numpy PCG64),
    and <= 512 evenly strided entries.  An error vector e moves the projection by ~ |e|_2 * N(0, 1), the sum by
    <= |e|_2 * sqrt(n), and shows up entry by entry in the sample.  Shared by tests/golden/make_golden.py (reference
    side) and tests/test_gpu_train.py (HIP side)."""
    g = np.asarray(grad.detach().cpu().numpy() if hasattr(grad, "detach") else grad, dtype=np.float64).reshape(-1)
    n = g.size
    vec = np.random.Generator(np.random.PCG64(zlib.crc32(name.encode()))).standard_normal(n)
    idx = np.unique(np.linspace(0, n - 1, min(n, 512)).astype(np.int64))
    return {"sum": np.float64(g.sum()), "l2": np.float64(np.sqrt((g * g).sum())), "max": np.float64(np.abs(g).max()),
            "proj": np.float64((g * vec).sum()), "sample": g[idx].astype(np.float32), "numel": np.int64(n)}


def check_grad_digests(g, grads, tol, dump=None):
    """Compare gradient tensors with the digests of a `*_full` fixture (tests/helpers.py::grad_digest): every strided
    sample entry within tol * max|g|, the name-seeded random projection within 4 tol * |g|_2 (an error vector e moves
    it by ~ |e|_2 N(0,1)), the L2 norm within tol.  Returns the worst relative errors (sample, projection, norm)."""
    worst = [0.0, 0.0, 0.0]
    report = {}
    for name in (str(n) for n in g["names"]):
        got = grad_digest(name, grads[name])
        ref = {f: g[f"grad.{f}.{name}"] for f in ("sum", "l2", "max", "proj", "sample", "numel")}
        assert int(got["numel"]) == int(ref["numel"]), name
        e_sample = float(np.abs(got["sample"].astype(np.float64) - ref["sample"].astype(np.float64)).max() / ref["max"])
        e_proj = float(abs(got["proj"] - ref["proj"]) / ref["l2"])
        e_l2 = float(abs(got["l2"] - ref["l2"]) / ref["l2"])
        report[name] = (e_sample, e_proj, e_l2)
        worst = [max(a, b) for a, b in zip(worst, (e_sample, e_proj, e_l2))]
    if dump:
        import json
        with open(dump, "w") as f:
            json.dump({"worst": worst, "per_tensor": report}, f, indent=1)
    for name, (e_sample, e_proj, e_l2) in report.items():
        assert e_sample <= tol and e_proj <= 4 * tol and e_l2 <= tol, (name, e_sample, e_proj, e_l2, tol)
    return worst
