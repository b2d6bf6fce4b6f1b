"""Data-side boundary types of the hot path (mirror of /root/reference/src/data): what crosses into
`training_step` / `validation_step` / `eval_nerf.py` -- the ray-batch container, the dataset classes the scripts and
`BaseModel.load_dataset` construct, the per-image ray cache, the NeRF-synthetic (Blender) reader and the LLFF scene reader (config 5).  COLMAP's own tooling and
the ScanNet reader are outside the scope table."""
from .data_helpers import DataBundle, batch_random_sampling, pose_spherical  # noqa: F401
from .datasets import (BlenderDataset, CachedRayDataset, CachingDataset, ColmapDataset, DatasetType,  # noqa: F401
                       SynthesizableDataset, convert_poses_to_rays)
