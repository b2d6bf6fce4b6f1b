"""File readers of the data feed (mirror of /root/reference/src/data/loaders): only the NeRF-synthetic (Blender) layout,
which the headline configs use; LLFF / COLMAP / ScanNet parsing is outside the scope table."""
from .load_blender import load_blender_data  # noqa: F401
