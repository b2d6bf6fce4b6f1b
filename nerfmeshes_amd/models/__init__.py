"""`models` package of the reference (/root/reference/src/models/__init__.py): same public names."""
from .model_base import BaseModel
from .model_buff import BuFFModel
from .model_nerf import NeRFModel
from .model_helpers import flatten_dict, intervals_to_ray_points, nest_dict
