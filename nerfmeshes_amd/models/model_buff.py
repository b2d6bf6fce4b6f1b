"""BuFFModel: single network + voxel-tree sampling (mirror of /root/reference/src/models/model_buff.py:12-73,166-170)."""
import torch

from ..nerf import RaySampleInterval, TreeSampling, models as nerf_models
from .model_base import BaseModel


class BuFFModel(BaseModel):
    def __init__(self, cfg, *args, **kwargs):
        super().__init__(cfg, *args, **kwargs)
        self.model = getattr(nerf_models, self.cfg.models.coarse_type)(**self.cfg.models.coarse)
        self.tree = TreeSampling(self.cfg, "cuda" if torch.cuda.is_available() else "cpu")
        self.sampler = RaySampleInterval(self.cfg.nerf.train.num_coarse)

    def get_model(self):
        return self.model

    def forward(self, x):
        ray_origins, ray_directions, (near, far) = x
        nerf_cfg = self.cfg.nerf.train if self.model.training else self.cfg.nerf.validation
        dev = self.model.layer1.weight.device
        ray_origins, ray_directions = ray_origins.to(dev), ray_directions.to(dev)
        ray_count = ray_directions.shape[0]
        uniform = self.sampler(nerf_cfg, ray_count, near, far)
        intervals, indices, mask = self.tree.batch_ray_voxel_intersect(ray_origins, ray_directions, near, far,
                                                                       samples_count=nerf_cfg.num_coarse)
        intervals[~mask] = uniform[~mask]                         # rays that miss every voxel (model_buff.py:53)
        radiance = self.model.hip().eval_rays(ray_origins, ray_directions, intervals)
        bundle = self.volume_renderer(radiance, intervals, ray_directions)
        if self.training:
            self.tree.ray_batch_integration(self.global_step, indices[mask], bundle.weights[mask],
                                            bundle.mask_weights[mask])
        return bundle

    def query(self, ray_batch):
        return self.forward(ray_batch)

    def on_save_checkpoint(self, checkpoint):
        checkpoint["tree"] = self.tree.serialize()

    def on_load_checkpoint(self, checkpoint):
        self.tree.deserialize(checkpoint["tree"])
