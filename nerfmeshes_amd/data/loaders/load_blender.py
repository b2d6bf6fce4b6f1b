"""NeRF-synthetic reader (mirror of /root/reference/src/data/loaders/load_blender.py:11-128): `transforms_<split>.json`
(`camera_angle_x`, `frames[].file_path / transform_matrix`) + one PNG per frame -> a `DataBundle` with
`ray_targets (N,H,W,3)` in [0,1], `poses (N,3,4)`, `hwf`, `size`; optional `<frame>_normal.png` / `<frame>_depth.exr`.
Images are decoded with Pillow (the reference uses imageio/cv2, not installed offline)."""
import json
import os
from pathlib import Path

import numpy as np
import torch

from ..data_helpers import DataBundle


def _read_png(path):
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im)


def _read_exr_depth(path):
    try:
        import Imath
        import OpenEXR
    except ImportError:
        return None
    f = OpenEXR.InputFile(str(path))
    dw = f.header()["dataWindow"]
    shape = (dw.max.y - dw.min.y + 1, dw.max.x - dw.min.x + 1)
    ch = "Z" if "Z" in f.header()["channels"] else "R"
    return np.frombuffer(f.channel(ch, Imath.PixelType(Imath.PixelType.FLOAT)), dtype=np.float32).reshape(shape).copy()


def load_blender_data(cfg, data_config):
    json_path = Path(data_config)
    if not json_path.exists():
        raise FileNotFoundError(
            f"{json_path}: no such transforms file. dataset.basedir must hold a NeRF-synthetic scene "
            "(transforms_train/val/test.json + images); alternatively enable dataset.caching.use_caching over an "
            "existing ray cache.")
    print(f"Reading from {json_path}...")
    with json_path.open("r") as fp:
        meta = json.load(fp)
    base = json_path.parent
    imgs, poses, depth, normals = [], [], [], []
    for frame in meta["frames"]:
        stem = base / frame["file_path"]
        png = stem.with_suffix(".png")
        if not png.exists():
            raise FileNotFoundError(f"{png}: image of frame {frame['file_path']!r} listed in {json_path.name} is missing")
        imgs.append(_read_png(png)[..., :3])                      # rgb only, as load_blender.py:42
        exr = Path(f"{stem}_depth.exr")
        if exr.exists():
            z = _read_exr_depth(exr)
            if z is not None:
                z[z == z.max(initial=0)] = cfg.dataset.empty      # background = the far plane of the render
                depth.append(z)
        npng = Path(f"{stem}_normal.png")
        if npng.exists():
            try:
                normals.append(_read_png(npng))
            except OSError:
                pass
        poses.append(np.array(frame["transform_matrix"])[:3, :4])
    size = len(imgs)
    print(f"Finished reading from {json_path} with {size} assets.")
    imgs = (np.array(imgs) / 255.0).astype(np.float32)
    depth = torch.from_numpy(np.array(depth).astype(np.float32)) if len(depth) == size and size else None
    if len(normals) == size and size:
        n = (np.array(normals) / 255.0).astype(np.float32)[..., :3]
        normals = torch.from_numpy(n / np.linalg.norm(n, axis=-1)[..., None])
    else:
        normals = None
    H, W = imgs[0].shape[:2]
    focal = 0.5 * W / np.tan(0.5 * float(meta["camera_angle_x"]))
    imgs = torch.from_numpy(imgs)
    red = cfg.dataset.reduced_resolution
    if red is not None and red > 1:
        H, W, focal = H // red, W // red, focal / red
        print(f"Using reduced resolution: {red} of size {W}x{H}")
        # area averaging (what cv2.INTER_AREA computes for an integer factor)
        imgs = torch.nn.functional.interpolate(imgs.permute(0, 3, 1, 2), size=(H, W), mode="area").permute(0, 2, 3, 1).contiguous()
    if cfg.dataset.white_background:
        # the reference blends over white with the LAST channel of the rgb image (the alpha plane was dropped above,
        # load_blender.py:42,113-114); reproduced as it is
        imgs = imgs * imgs[..., -1:] + (1.0 - imgs[..., -1:])
    return DataBundle(ray_targets=imgs, target_depth=depth, target_normals=normals,
                      poses=torch.from_numpy(np.array(poses).astype(np.float32)), hwf=(H, W, focal), size=size)
