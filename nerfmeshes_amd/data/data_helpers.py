"""`DataBundle` with the reference's field names and methods (/root/reference/src/data/data_helpers.py:83-171):
the dict <-> bundle round trip the DataLoader performs (`serialize` / `deserialize`), `to_ray_batch`, `to`,
indexing by ray, `ndc`.  `batch_random_sampling` (:42-54) picks the training rays of an image."""
from dataclasses import dataclass, fields

import torch

from ..nerf.nerf_helpers import ndc_rays
from ..synthetic import pose_spherical  # noqa: F401  (data_helpers.py:33-39, same matrix)

_FLAT_VIEWS = {"ray_origins": (-1, 3), "ray_directions": (-1, 3), "ray_bounds": (2,), "ray_targets": (-1, 3),
               "target_depth": (-1,)}


@dataclass
class DataBundle:
    ray_origins: torch.Tensor = None
    ray_directions: torch.Tensor = None
    ray_targets: torch.Tensor = None
    ray_bounds: torch.Tensor = None
    target_depth: torch.Tensor = None
    target_normals: torch.Tensor = None
    poses: torch.Tensor = None
    size: int = -1
    hwf: tuple = None

    def _names(self):
        return [f.name for f in fields(self)]

    def __iter__(self):
        return iter(tuple(getattr(self, n) for n in self._names()))

    def __getitem__(self, key):
        if isinstance(key, int):           # one ray of a per-ray bundle
            picked = DataBundle()
            for n in self._names():
                v = getattr(self, n)
                per_ray = isinstance(v, torch.Tensor) and v.shape[0] == self.size
                setattr(picked, n, v[key] if per_ray else v)
            return picked
        if isinstance(key, tuple):
            return iter([getattr(self, k) for k in key])
        return getattr(self, key)

    @staticmethod
    def deserialize(mapping):
        bundle = DataBundle()
        for n in bundle._names():
            if n in mapping:
                setattr(bundle, n, mapping[n])
        return bundle

    def serialize(self, filters):
        return {n: getattr(self, n) for n in self._names() if n in filters and getattr(self, n) is not None}

    def apply(self, func, names):
        present = [n for n in names if getattr(self, n) is not None]
        results = func([getattr(self, n) for n in present])
        out = DataBundle(**{n: getattr(self, n) for n in self._names()})
        for n, v in zip(present, results):
            setattr(out, n, v)
        return out

    def to_ray_batch(self):
        """Drop the DataLoader's batch dimension: origins/directions/targets (-1,3), bounds (2,), depth (-1,)."""
        for n, shape in _FLAT_VIEWS.items():
            v = getattr(self, n)
            if v is not None:
                setattr(self, n, v.view(*shape))
        return self

    def to(self, device):
        for n in self._names():
            v = getattr(self, n)
            if isinstance(v, torch.Tensor):
                setattr(self, n, v.to(device))
        return self

    def ndc(self):
        self.ray_origins, self.ray_directions = ndc_rays(*self.hwf, 1.0, self.ray_origins[None, None, :],
                                                         self.ray_directions)
        return self


def batch_random_sampling(cfg, coords, ray_bundle):
    """cfg.nerf.train.num_random_rays random pixels of one image: coords (H*W, 2) integer (row, col) pairs,
    ray_bundle a tuple of (H, W, ...) tensors (or None)."""
    chosen = coords[torch.randperm(coords.shape[0], device=coords.device)[:cfg.nerf.train.num_random_rays]]
    rows, cols = chosen[:, 0], chosen[:, 1]
    return tuple(None if item is None else item[rows, cols, ...] for item in ray_bundle)
