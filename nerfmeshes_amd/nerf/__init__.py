"""`nerf` package of the reference (/root/reference/src/nerf/__init__.py), same public names."""
from .cfgnode import CfgNode
from .tree import Node, TreeSampling
from .nerf_helpers import (POINT_GROUND_TRUTH, POINT_OUT_FALSE_SURFACE, POINT_OUT_FALSE_VOID, POINT_OUT_TRUE, batchify,
                           cast_to_disparity_image, cast_to_image, cast_to_pil_image, comp_depth, create_point_cloud,
                           cumprod_exclusive, export_obj, export_point_cloud, get_point_clouds, get_ray_bundle, img2mse,
                           meshgrid_xy, mse2psnr, ndc_rays)
from .modules import OutputBundle, PositionalEncoding, RaySampleInterval, SamplePDF, VolumeRenderer
from .models import FlexibleNeRFModel
from . import models, modules, nerf_helpers, tree
