"""Golden outputs of the UNMODIFIED reference's FlexibleNeRFModel(use_viewdirs=False) (/root/reference/src/nerf/models.py:
52-55, 77-79: the trunk ends in fc_out, colours through a sigmoid, density raw) on seeded weights and points.

    python tests/golden/make_flat_golden.py        # container only (needs /root/reference); writes mlp_flat_points.npz

The oracle's use_viewdirs=False branch and the HIP kernels' mode 2 are checked against this file
(tests/test_oracle_golden.py, tests/test_gpu_parity.py)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
import ref_import  # noqa: E402

from nerfmeshes_amd import synthetic as S  # noqa: E402

CASES = {
    "a": dict(num_layers=8, hidden_size=256, skip_step=4, num_encoding_fn_xyz=10, num_encoding_fn_dir=4),
    "b": dict(num_layers=4, hidden_size=128, skip_step=4, num_encoding_fn_xyz=6, num_encoding_fn_dir=4),
    "c": dict(num_layers=6, hidden_size=64, skip_step=2, num_encoding_fn_xyz=6, num_encoding_fn_dir=4),
}
SEED, GAIN, BIAS, POINTS = 31, 40.0, 1.0, 384


def main():
    nerf, _ = ref_import.load()
    out = {"seed": SEED, "gain": GAIN, "bias": BIAS}
    g = torch.Generator().manual_seed(12)
    pts = (torch.rand(POINTS, 3, generator=g) * 2 - 1) * torch.tensor([6.0, 1.2, 3.0])
    out["points"] = pts.numpy()
    for tag, kw in CASES.items():
        net = nerf.FlexibleNeRFModel(use_viewdirs=False, **kw).eval()
        w = S.make_mlp_weights(SEED, density_gain=GAIN, density_bias=BIAS, use_viewdirs=False, **kw)
        sd = net.state_dict()
        for k, v in w.items():
            assert tuple(sd[k].shape) == v.shape, (k, sd[k].shape, v.shape)
            sd[k] = torch.from_numpy(v)
        net.load_state_dict(sd)
        with torch.no_grad():
            rad = net(pts.clone(), None)
        out["radiance_" + tag] = rad.numpy()
        for k, v in kw.items():
            out[f"{k}_{tag}"] = v
        print(tag, kw, rad.shape, float(rad[:, 3].abs().max()))
    np.savez_compressed(os.path.join(HERE, "mlp_flat_points.npz"), **out)


if __name__ == "__main__":
    main()
