"""Calibrate the smooth synthetic scene (nerfmeshes_amd.synthetic.SMOOTH_SCENES) for a network shape: statistics of the
raw fc_alpha output of the band-limited seeded draw along lego-orbit rays.  The constants it prints are hard-coded in
synthetic.py so that every host regenerates bit-identical weights.

    python tests/golden/calibrate_scene.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from nerfmeshes_amd import synthetic as S  # noqa: E402
from oracle import nerf_oracle as O  # noqa: E402

SHAPES = {
    "fern_8x128": dict(num_layers=8, hidden_size=128, skip_step=4, num_encoding_fn_xyz=10, num_encoding_fn_dir=4),
    "tiny_4x64": dict(num_layers=4, hidden_size=64, skip_step=4, num_encoding_fn_xyz=6, num_encoding_fn_dir=4),
}


def permute_hidden_units(w, H, L, seed=0):
    """The same function with the hidden units of every trunk layer permuted: mathematically neutral, numerically a
    different summation order in every dot product over a hidden layer."""
    rng = np.random.default_rng(seed)
    w = {k: v.copy() for k, v in w.items()}
    names = ["layer1"] + [f"layers_xyz.{i}" for i in range(L - 1)]
    for li, n in enumerate(names):
        p = rng.permutation(H)
        w[n + ".weight"], w[n + ".bias"] = w[n + ".weight"][p], w[n + ".bias"][p]
        for nxt in ([names[li + 1]] if li + 1 < len(names) else ["fc_feat", "fc_alpha"]):
            w[nxt + ".weight"][:, :H] = w[nxt + ".weight"][:, :H][:, p]
    return w


def self_noise(name, rays=8192):
    """|dPSNR| and max |d rgb| of the reference path against ITSELF (three hidden-unit permutations) on the rays of the
    strict-bar fixture of scene `name`: the floor below which no implementation can be compared."""
    from oracle import parity
    w, kw = S.make_smooth_scene_weights(name)
    spec = O.MLPSpec(**kw)
    rs = O.RenderSpec() if name != "tiny_4x64" else O.RenderSpec(num_coarse=32, num_fine=0)
    o, d = O.get_ray_bundle(800, 800, S.LEGO_FOCAL_800, torch.from_numpy(S.orbit_poses(4)[0]))
    d = d.reshape(-1, 3)[torch.arange(rays) * (640000 // rays) + 7]

    def render(wt):
        with torch.no_grad():
            outs = [O.render(wt, wt if rs.num_fine else None, spec, spec, rs, o[None], d[s:s + 2048], 2.0, 6.0) for s in range(0, rays, 2048)]
        return torch.cat([(f if f is not None else c)["rgb_map"] for c, f in outs])

    a = render(w)
    out = []
    for seed in (0, 1, 2):
        p = parity.psnr_parity(render(permute_hidden_units(w, kw["hidden_size"], kw["num_layers"], seed)), a, chunk=2048)
        out.append((p["abs_dpsnr_db"], p["max_abs_drgb"], p["rays_over_1e-4"]))
    return out


def main():
    for name, kw in SHAPES.items():
        w = S.band_limit(S.make_mlp_weights(S.SCENE_SEED, **kw), S.SCENE_DECAY, **kw)
        o, d = O.get_ray_bundle(800, 800, S.LEGO_FOCAL_800, torch.from_numpy(S.orbit_poses(4)[0]))
        d = d.reshape(-1, 3)[7::157][:4096]
        t = torch.linspace(2.0, 6.0, 64)[None, :].expand(d.shape[0], -1)
        pts = (o[None, None, :] + d[:, None, :] * t[..., None]).reshape(-1, 3)
        dirs = d[:, None, :].expand(-1, 64, -1).reshape(-1, 3)
        out = O.mlp_forward({k: torch.from_numpy(v) for k, v in w.items()}, O.MLPSpec(**kw), pts, dirs)
        raw = out[..., 3].double()
        print(f'    "{name}": ({float(raw.mean()):.6f}, {float(raw.std()):.6f}),')
    torch.set_num_threads(min(16, torch.get_num_threads()))
    for name in SHAPES:
        print(name, "self-noise of the reference path (|dPSNR| dB, max |d rgb|, rays over 1e-4) x 3 permutations:", self_noise(name))


if __name__ == "__main__":
    main()
