"""Dataset classes of the data feed, with the reference's names (SURVEY.md 8(f) rank 4;
/root/reference/src/data/datasets.py:22-328): `DatasetType`, `SynthesizableDataset`, `CachingDataset`, `BlenderDataset`,
`ColmapDataset` -- what `eval_nerf.py:27-31` and `BaseModel.load_dataset` (`models/model_base.py:105-117`) construct.

A sample is ONE image: `__getitem__` returns `DataBundle.serialize(filters)` -- the whole image's rays, or
`nerf.train.num_random_rays` random ones of it for the TRAIN split.  Two storage modes, as in the reference:

* `dataset.caching.use_caching`: one file per image, `<cache_dir>/<train|val|test>/NNNN.data` =
  `torch.save(bundle.serialize(filters))` (datasets.py:248-283).  A cache written by the reference feeds this class and
  vice versa (`tests/golden/ref_cache/` was written by the reference's own `CachingDataset`).  A missing cache is built
  from `load_dataset()`, the rays generated on the GPU (`nm_ray_bundle`, `nm_ndc_rays`).
* otherwise the whole split lives in memory with rays for every pose.

`load_dataset()` is the file reader.  `BlenderDataset` reads the NeRF-synthetic layout (`transforms_<split>.json` + PNGs,
`loaders/load_blender.py`), `ColmapDataset` the LLFF layout (`poses_bounds.npy` + `images_<factor>/`, `loaders/load_llff.py`:
recentred / spherified poses, bounds scaled to the nearest depth, hold-out split) -- both also work from an existing ray cache.
"""
import glob
import os
import time
from enum import Enum
from pathlib import Path

import numpy as np
import torch
from torch.utils.data import Dataset

from ..nerf.nerf_helpers import get_ray_bundle, meshgrid_xy
from .data_helpers import DataBundle, batch_random_sampling, pose_spherical

FILTERS = ["ray_origins", "ray_directions", "ray_targets", "ray_bounds", "target_depth", "size", "hwf"]
PER_PIXEL = ["ray_directions", "ray_targets", "target_depth", "target_normals"]


class DatasetType(Enum):
    TRAIN = "train"
    TEST = "test"
    VALIDATION = "val"


def convert_poses_to_rays(poses, H, W, focal):
    """datasets.py:47-59: rays of every pose -> (origins (N,3), directions (N,H,W,3)); generated on the GPU."""
    origins, directions = zip(*(get_ray_bundle(H, W, focal, pose) for pose in poses))
    return torch.stack(origins, 0), torch.stack(directions, 0)


class SynthesizableDataset(Dataset):
    """datasets.py:82-133: `synthesis()` replaces the images by a 360-degree orbit of novel views (no targets)."""

    STEP_SIZE = 3

    def __init__(self):
        super().__init__()
        self.synthetic_bundle = None

    def synthesis(self):
        print("Synthesizing dataset...")
        angles = np.linspace(-270, 90, 360 // SynthesizableDataset.STEP_SIZE, endpoint=False)
        poses = torch.stack([torch.from_numpy(pose_spherical(a, -30.0, 4.0)) for a in angles], 0)
        hwf = self.data_bundle.hwf if self.data_bundle is not None else self.hwf
        self.synthetic_bundle = DataBundle(poses=poses, ray_bounds=self.ray_bounds, hwf=hwf, size=len(poses))
        o, d = convert_poses_to_rays(poses, *hwf)
        self.synthetic_bundle.ray_origins, self.synthetic_bundle.ray_directions = o.cpu(), d.cpu()


class CachingDataset(SynthesizableDataset):
    """datasets.py:136-291."""

    def __init__(self, cfg, type):
        super().__init__()
        if not isinstance(type, DatasetType):
            raise AssertionError(f"Invalid dataset type {type} expected {[t.name for t in DatasetType]}")
        self.cfg, self.type = cfg, type
        self.shuffle = True
        self.data_bundle = None
        self.hwf = None
        self.coords = None
        self.paths = []
        self.filters = list(FILTERS)
        self.ray_bounds = torch.tensor([cfg.dataset.near, cfg.dataset.far]).float()
        self.num_random_rays = cfg.nerf.train.num_random_rays
        self.path = os.path.join(cfg.dataset.caching.cache_dir, type.value)
        self.device = "cuda" if torch.cuda.is_available() else "cpu"
        start = time.time()
        if cfg.dataset.caching.use_caching:
            existed = os.path.exists(self.path)
            if not existed:
                print(f"The path ${self.path} does not exist, creating one...")
                os.makedirs(self.path, exist_ok=True)
            if cfg.dataset.caching.override_caching or not existed:
                print(f"{'Overriding' if existed else 'Creating'} the cached dataset to {self.path}...")
                self.cache_dataset()
            else:
                print(f"Using existent cached dataset from {self.path}...")
            self.refresh()
            if not self.paths and existed:
                print(f"The previous cached dataset is corrupted in {self.path}, overriding it...")
                self.cache_dataset()
                self.refresh()
            assert len(self.paths) > 0, "There is a critical issue when caching the dataset"
            size = len(self.paths)
            print(f"Using cached dataset in {time.time() - start}s seconds with {size} assets...")
        else:
            self.data_bundle = self.load_dataset()
            self.init_sampling(self.data_bundle.hwf)
            o, d = convert_poses_to_rays(self.data_bundle.poses, *self.data_bundle.hwf)
            self.data_bundle.ray_origins, self.data_bundle.ray_directions = o, d
            if cfg.dataset.use_ndc:
                self._ndc_all(self.data_bundle)            # on the GPU, where the rays were generated; leaves host tensors
            else:
                self.data_bundle.ray_origins, self.data_bundle.ray_directions = o.cpu(), d.cpu()
            print(f"Load whole dataset into the memory {time.time() - start}s seconds...")

    @staticmethod
    def _ndc_all(bundle):
        """`DataBundle.ndc` per image (the reference calls it on the stacked bundle, which only broadcasts for one
        image): origins become per-pixel (N,H,W,3)."""
        pairs = [DataBundle(ray_origins=o, ray_directions=d, hwf=bundle.hwf).ndc()
                 for o, d in zip(bundle.ray_origins, bundle.ray_directions)]
        bundle.ray_origins = torch.stack([p.ray_origins.reshape(p.ray_directions.shape) for p in pairs], 0).cpu()
        bundle.ray_directions = torch.stack([p.ray_directions for p in pairs], 0).cpu()

    def refresh(self):
        """Re-scan the cache directory (sorted: image order = file-name order)."""
        self.paths = sorted(glob.glob(os.path.join(self.path, "*.data")))
        if self.paths:
            self.hwf = tuple(torch.load(self.paths[0], weights_only=False)["hwf"])
            self.init_sampling(self.hwf)

    def __len__(self):
        if self.synthetic_bundle is not None:
            return self.synthetic_bundle.size
        return len(self.paths) if self.cfg.dataset.caching.use_caching else self.data_bundle.size

    def __getitem__(self, idx):
        if self.synthetic_bundle is not None:          # novel views take precedence over either storage mode
            bundle = self.synthetic_bundle[idx]
        elif self.cfg.dataset.caching.use_caching:
            bundle = DataBundle.deserialize(torch.load(self.paths[idx], weights_only=False))
        else:
            bundle = self.data_bundle[idx]
        if self.type == DatasetType.TRAIN:                          # datasets.py:227-234
            names = (["ray_origins"] if self.cfg.dataset.use_ndc else []) + PER_PIXEL
            bundle = bundle.apply(lambda items: batch_random_sampling(self.cfg, self.coords, items), names)
        return bundle.serialize(self.filters)

    def init_sampling(self, hwf):
        """datasets.py:238-246: the (H*W, 2) pixel coordinates the random rays are drawn from."""
        height, width = int(hwf[0]), int(hwf[1])
        self.coords = torch.stack(meshgrid_xy(torch.arange(height), torch.arange(width)), dim=-1).reshape(-1, 2)

    def save_dataset(self, bundle, img_idx, batch_idx=-1):
        """datasets.py:248-262: cache one image's rays (`bundle` with (H,W,3) directions / targets)."""
        if batch_idx != -1:
            raise NotImplementedError
        os.makedirs(self.path, exist_ok=True)
        torch.save(bundle.to("cpu").serialize(self.filters), os.path.join(self.path, str(img_idx).zfill(4) + ".data"))

    write_view = save_dataset

    def cache_dataset(self):
        """datasets.py:264-283: read the split once, generate every image's rays on the GPU, one file per image."""
        bundle = self.load_dataset()
        self.init_sampling(bundle.hwf)
        if not (self.cfg.dataset.caching.sample_all or self.type == DatasetType.VALIDATION):
            raise NotImplementedError
        for img_idx in range(bundle.size):
            sample = bundle[img_idx]
            sample.ray_origins, sample.ray_directions = get_ray_bundle(*sample.hwf, sample.poses)
            if self.cfg.dataset.use_ndc:
                sample.ndc()
            self.save_dataset(sample, img_idx)

    @property
    def dataset_path(self):
        return Path(self.cfg.dataset.basedir)

    def load_dataset(self):
        raise NotImplementedError(f"{type(self).__name__}.load_dataset: no reader for this dataset type")


class BlenderDataset(CachingDataset):
    """datasets.py:294-314: NeRF-synthetic scenes; one sample = one full image."""

    def __init__(self, cfg, type=DatasetType.TRAIN):
        super().__init__(cfg, type)
        print("Loading Blender Data...")

    @property
    def dataset_path(self):
        return Path(self.cfg.dataset.basedir) / f"transforms_{self.type.value}.json"

    def load_dataset(self):
        from .loaders.load_blender import load_blender_data
        bundle = load_blender_data(self.cfg, self.dataset_path)
        if bundle.ray_bounds is None:
            bundle.ray_bounds = self.ray_bounds
        return bundle


class ColmapDataset(CachingDataset):
    """datasets.py:317-357: LLFF / COLMAP forward-facing (or, `spherify`, inward-facing) captures; one sample = one full image.
    `dataset.basedir` holds poses_bounds.npy + images[_<llff_downsample_factor>]/ (loaders/load_llff.py); with
    `dataset.caching.use_caching` an existing ray cache is served as it is, whatever wrote it."""

    def __init__(self, cfg, spherify=True, type=DatasetType.TRAIN):
        self.downscale_factor = cfg.dataset.llff_downsample_factor
        self.spherify = spherify
        super().__init__(cfg, type)
        print("Loading Colmap Data...")

    def load_dataset(self):
        from .loaders.load_llff import load_llff_data
        if not (Path(self.dataset_path) / "poses_bounds.npy").exists():
            raise FileNotFoundError(
                f"{Path(self.dataset_path) / 'poses_bounds.npy'}: dataset.basedir must hold an LLFF scene (poses_bounds.npy + "
                f"images_{self.downscale_factor}/); alternatively point dataset.caching.cache_dir at a ray cache "
                f"(use_caching: True; looked in {self.path}) -- the files the reference's CachingDataset writes are read as they are.")
        images, pose_mats, bounds, _render_poses, i_test = load_llff_data(str(self.dataset_path), factor=self.downscale_factor,
                                                                           spherify_poses=self.spherify)
        # hold out every llff_hold_step-th view for validation, or only the one closest to the average pose (datasets.py:330-338)
        every = self.cfg.dataset.llff_hold_step
        held = np.arange(images.shape[0])[::every] if every > 0 else np.array([i_test])
        kept = np.array([i for i in np.arange(images.shape[0]) if i not in held])
        pick = kept if self.type == DatasetType.TRAIN else held
        pose_mats = torch.from_numpy(pose_mats[pick, ...])
        return DataBundle(ray_targets=torch.from_numpy(images[pick, ...]), ray_bounds=torch.from_numpy(bounds[pick, ...]),
                          poses=pose_mats[:, :3, :4], hwf=tuple(pose_mats[0, :3, -1].long().tolist()), size=len(pick))


class CachedRayDataset(CachingDataset):
    """A dataset over a ray cache and nothing else, whatever reader produced the files: never calls `load_dataset`,
    ignores `dataset.caching.use_caching`, and may start empty (`write_view` fills it, `refresh` re-scans)."""

    def __init__(self, cfg, type=DatasetType.TRAIN):
        SynthesizableDataset.__init__(self)
        if not isinstance(type, DatasetType):
            raise ValueError(f"Invalid dataset type {type} expected {[t.name for t in DatasetType]}")
        self.cfg, self.type = cfg, type
        self.data_bundle, self.hwf, self.coords, self.paths = None, None, None, []
        self.filters = list(FILTERS)
        self.ray_bounds = torch.tensor([cfg.dataset.near, cfg.dataset.far]).float()
        self.path = os.path.join(cfg.dataset.caching.cache_dir, type.value)
        self.refresh()

    def __len__(self):
        return self.synthetic_bundle.size if self.synthetic_bundle is not None else len(self.paths)

    def __getitem__(self, idx):
        if self.synthetic_bundle is None and not self.paths:
            raise IndexError(f"empty ray cache {self.path}")
        bundle = self.synthetic_bundle[idx] if self.synthetic_bundle is not None else \
            DataBundle.deserialize(torch.load(self.paths[idx], weights_only=False))
        if self.type == DatasetType.TRAIN:
            names = (["ray_origins"] if self.cfg.dataset.use_ndc else []) + PER_PIXEL
            bundle = bundle.apply(lambda items: batch_random_sampling(self.cfg, self.coords, items), names)
        return bundle.serialize(self.filters)
