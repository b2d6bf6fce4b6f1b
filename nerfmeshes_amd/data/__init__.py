"""Data-side boundary types of the hot path (mirror of /root/reference/src/data): only what crosses into
`training_step` / `validation_step` / `eval_nerf.py` -- the ray-batch container and the per-image ray cache.  Dataset
readers (Blender / COLMAP image loading) are outside the scope table."""
from .data_helpers import DataBundle, batch_random_sampling, pose_spherical  # noqa: F401
from .datasets import CachedRayDataset, DatasetType  # noqa: F401
