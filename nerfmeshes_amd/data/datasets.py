"""The ray cache of the data feed (/root/reference/src/data/datasets.py:136-283).

The reference pre-computes the rays of every image once and stores one file per image,
`<dataset.caching.cache_dir>/<train|val|test>/NNNN.data` = `torch.save(bundle.serialize(filters))` with
filters = ray_origins, ray_directions, ray_targets, ray_bounds, target_depth, size, hwf.  `CachedRayDataset` reads
and writes exactly those files (so a cache made by the reference feeds this training loop and vice versa) and draws
the `nerf.train.num_random_rays` training rays of an image the way `CachingDataset.__getitem__` does.  Image / COLMAP
readers (`load_dataset` of the Blender / COLMAP subclasses) stay out of scope; `write_view` is the hook they would
call, and ray generation for it runs on the GPU (`hip_ops.ray_bundle`).
"""
import glob
import os
from enum import Enum

import torch
from torch.utils.data import Dataset

from ..nerf.nerf_helpers import meshgrid_xy
from .data_helpers import DataBundle, batch_random_sampling

FILTERS = ["ray_origins", "ray_directions", "ray_targets", "ray_bounds", "target_depth", "size", "hwf"]
PER_PIXEL = ["ray_directions", "ray_targets", "target_depth", "target_normals"]


class DatasetType(Enum):
    TRAIN = "train"
    TEST = "test"
    VALIDATION = "val"


class CachedRayDataset(Dataset):
    def __init__(self, cfg, type=DatasetType.TRAIN):
        if not isinstance(type, DatasetType):
            raise ValueError(f"Invalid dataset type {type} expected {[t.name for t in DatasetType]}")
        self.cfg, self.type = cfg, type
        self.filters = list(FILTERS)
        self.path = os.path.join(cfg.dataset.caching.cache_dir, type.value)
        self.coords = None
        self.refresh()

    def refresh(self):
        self.paths = sorted(glob.glob(os.path.join(self.path, "*.data")))
        if self.paths:
            self.init_sampling(torch.load(self.paths[0], weights_only=False)["hwf"])

    def init_sampling(self, hwf):
        """datasets.py:238-246: the (H*W, 2) pixel coordinates the random rays are drawn from."""
        height, width = int(hwf[0]), int(hwf[1])
        self.coords = torch.stack(meshgrid_xy(torch.arange(height), torch.arange(width)), dim=-1).reshape(-1, 2)

    def write_view(self, bundle, img_idx):
        """datasets.py:248-262: cache one image's rays (`bundle` with (H,W,3) directions / targets)."""
        os.makedirs(self.path, exist_ok=True)
        torch.save(bundle.to("cpu").serialize(self.filters), os.path.join(self.path, str(img_idx).zfill(4) + ".data"))

    save_dataset = write_view

    def __len__(self):
        return len(self.paths)

    def __getitem__(self, idx):
        bundle = DataBundle.deserialize(torch.load(self.paths[idx], weights_only=False))
        if self.type == DatasetType.TRAIN:                       # datasets.py:227-234
            names = (["ray_origins"] if self.cfg.dataset.use_ndc else []) + PER_PIXEL
            bundle = bundle.apply(lambda items: batch_random_sampling(self.cfg, self.coords, items), names)
        return bundle.serialize(self.filters)
