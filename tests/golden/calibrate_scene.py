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


if __name__ == "__main__":
    main()
