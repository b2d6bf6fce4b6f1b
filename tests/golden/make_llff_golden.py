"""tests/golden/llff_scene.npz: what the UNMODIFIED reference reader (/root/reference/src/data/loaders/load_llff.py: `load_llff_data`)
returns for a small synthetic LLFF scene, forward-facing (spiral render path) and spherified (circle), next to the scene itself --
poses_bounds.npy as an array and the decoded images_4/ PNGs -- so that the test can write the same folder again and hold
nerfmeshes_amd.data.loaders.load_llff against it without the reference being present.

    python tests/golden/make_llff_golden.py        (in the build container: needs /root/reference)

The reference module is executed as it is; only `imageio.imread` (not installed offline) is served by Pillow, which decodes the
same PNG bytes to the same uint8 arrays.  images_4/ is written directly (the reference would shell out to ImageMagick's mogrify
to create it from images/; an existing folder is used as it is, load_llff.py:12-23)."""
import importlib.util
import os
import sys
import tempfile
import types

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("NERFMESHES_REFERENCE", "/root/reference") + "/src/data/loaders/load_llff.py"


def make_scene(seed=7, views=9, height=6, width=8, factor=4):
    """A plausible hand-held capture: cameras on a jittered arc in front of the scene, looking roughly at a common point.
    -> (poses_bounds (views, 17) fp64, full-size images uint8, down-scaled images uint8)."""
    rng = np.random.default_rng(seed)
    rows = []
    for i in range(views):
        a = (i - (views - 1) / 2) * 0.12
        position = np.array([1.8 * np.sin(a), 0.15 * np.cos(3 * a), 0.3 * (1 - np.cos(a))]) + 0.03 * rng.standard_normal(3)
        back = position - np.array([0.0, 0.0, -4.0]) + 0.05 * rng.standard_normal(3)
        back /= np.linalg.norm(back)
        right = np.cross([0.0, 1.0, 0.0], back)
        right /= np.linalg.norm(right)
        up = np.cross(back, right)
        # LLFF's file order of the rotation columns: (down, right, back)
        block = np.stack([-up, right, back, position, [height * factor, width * factor, 30.0 * factor]], 1)
        rows.append(np.concatenate([block.ravel(), [2.1 + 0.2 * rng.random(), 9.0 + 3.0 * rng.random()]]))
    full = rng.integers(0, 256, (views, height * factor, width * factor, 3), dtype=np.uint8)
    small = rng.integers(0, 256, (views, height, width, 3), dtype=np.uint8)
    return np.array(rows), full, small


def write_scene(folder, table, full, small, factor=4):
    np.save(os.path.join(folder, "poses_bounds.npy"), table)
    for name, stack in (("images", full), (f"images_{factor}", small)):
        os.makedirs(os.path.join(folder, name), exist_ok=True)
        for i, img in enumerate(stack):
            Image.fromarray(img).save(os.path.join(folder, name, f"view_{i:03d}.png"))


def reference_reader():
    imageio = types.ModuleType("imageio")
    imageio.imread = lambda path, **kw: np.asarray(Image.open(path))
    sys.modules["imageio"] = imageio
    spec = importlib.util.spec_from_file_location("reference_load_llff", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    table, full, small = make_scene()
    ref = reference_reader()
    out = {"poses_bounds": table, "images_full_shape": np.array(full.shape), "images_4": small}
    with tempfile.TemporaryDirectory() as folder:
        write_scene(folder, table, full, small)
        for tag, kw in (("forward", dict(spherify=False)), ("spherify", dict(spherify=True))):
            images, poses, bds, render_poses, i_test = ref.load_llff_data(folder, factor=4, **kw)
            out.update({f"{tag}_images": images, f"{tag}_poses": poses, f"{tag}_bounds": bds, f"{tag}_render_poses": render_poses,
                        f"{tag}_i_test": np.array(i_test)})
    np.savez_compressed(os.path.join(HERE, "llff_scene.npz"), **out)
    print({k: (v.shape, str(v.dtype)) for k, v in out.items()})
