"""Runs under /opt/conda/bin/python3.9 (scikit-image 0.18.3): the black box behind
/root/reference/src/mesh_nerf.py:79.  Usage: python3.9 skimage_bridge.py in.npz out.npz
in.npz: vol_<i> (fp32 3-D), iso_<i> (float64).  out.npz: verts_<i>, faces_<i>, normals_<i>, values_<i>,
or err_<i> (the exception text) when skimage raises."""
import sys

import numpy as np
from skimage import measure

src = np.load(sys.argv[1])
out = {}
i = 0
while f"vol_{i}" in src.files:
    try:
        v, f, n, val = measure.marching_cubes(src[f"vol_{i}"], float(src[f"iso_{i}"]))
        out[f"verts_{i}"], out[f"faces_{i}"], out[f"normals_{i}"], out[f"values_{i}"] = v, f, n, val
    except Exception as e:  # noqa: BLE001
        out[f"err_{i}"] = np.array(f"{type(e).__name__}: {e}")
    i += 1
np.savez(sys.argv[2], **out)
