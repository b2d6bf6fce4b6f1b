"""Golden outputs of the UNMODIFIED reference's FlexibleNeRFModel (/root/reference/src/nerf/models.py:5-80) for network shapes NO
shipped config uses -- what the generic-shape kernel family (nerfmeshes_amd/csrc/mlp_device_g.h) serves: wide and odd hidden
sizes, unusual encoding lengths, include_input_* off, linear frequency sampling, with and without view directions.

    python tests/golden/make_generic_golden.py        # container only (needs /root/reference); writes mlp_generic_points.npz

The oracle is checked against this file bit for bit (tests/test_oracle_golden.py), the HIP kernels at 2e-5
(tests/test_gpu_generic.py)."""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
import ref_import  # noqa: E402

from nerfmeshes_amd import synthetic as S  # noqa: E402

BASE = dict(num_layers=8, hidden_size=256, skip_step=4, num_encoding_fn_xyz=10, num_encoding_fn_dir=4, include_input_xyz=True,
            include_input_dir=True, log_sampling_xyz=True, log_sampling_dir=True, use_viewdirs=True)
CASES = {
    "wide": dict(hidden_size=512),
    "odd": dict(num_layers=4, hidden_size=100, num_encoding_fn_xyz=7, num_encoding_fn_dir=1),
    "noinput": dict(num_layers=6, hidden_size=192, skip_step=3, include_input_xyz=False, include_input_dir=False),
    "linear": dict(num_layers=3, hidden_size=64, num_encoding_fn_xyz=3, num_encoding_fn_dir=5, log_sampling_xyz=False, log_sampling_dir=False),
    "long": dict(num_layers=4, hidden_size=272, num_encoding_fn_xyz=15, num_encoding_fn_dir=15),
    "rawdir": dict(num_layers=4, hidden_size=384, num_encoding_fn_dir=0),
    "flat": dict(num_layers=4, hidden_size=400, use_viewdirs=False),
}
SEED, GAIN, BIAS, POINTS = 37, 40.0, 1.0, 320


def main():
    nerf, _ = ref_import.load()
    out = {"seed": SEED, "gain": GAIN, "bias": BIAS}
    g = torch.Generator().manual_seed(13)
    pts = (torch.rand(POINTS, 3, generator=g) * 2 - 1) * torch.tensor([4.0, 1.2, 3.0])
    dirs = torch.nn.functional.normalize(torch.randn(POINTS, 3, generator=g), dim=-1)
    out["points"], out["directions"] = pts.numpy(), dirs.numpy()
    for tag, over in CASES.items():
        kw = dict(BASE, **over)
        net = nerf.FlexibleNeRFModel(**kw).eval()
        w = S.make_mlp_weights(SEED, density_gain=GAIN, density_bias=BIAS, **kw)
        sd = net.state_dict()
        for k, v in w.items():
            assert tuple(sd[k].shape) == v.shape, (k, sd[k].shape, v.shape)
            sd[k] = torch.from_numpy(v)
        net.load_state_dict(sd)
        with torch.no_grad():
            rad = net(pts.clone(), dirs.clone())
        out["radiance_" + tag] = rad.numpy()
        out["kwargs_" + tag] = json.dumps(kw)
        print(tag, over, rad.shape, float(rad[:, 3].abs().max()))
    np.savez_compressed(os.path.join(HERE, "mlp_generic_points.npz"), **out)


if __name__ == "__main__":
    main()
