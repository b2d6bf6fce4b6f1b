"""BuFFModel: single network + voxel-tree sampling (mirror of /root/reference/src/models/model_buff.py:12-73,166-170)."""
import torch

from .. import train_ops
from ..data import DataBundle
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
        self.tree.training = self.training                        # selects the reference's own voxel ids where they are consumed
        intervals, indices, mask = self.tree.batch_ray_voxel_intersect(ray_origins, ray_directions, near, far,
                                                                       samples_count=nerf_cfg.num_coarse)
        intervals[~mask] = uniform[~mask]                         # rays that miss every voxel (model_buff.py:53)
        if self.model.needs_grad():                               # training: differentiable kernels (train_ops)
            radiance = train_ops.mlp_rays(self.model, ray_origins, ray_directions, intervals)
        else:
            radiance = self.model.hip().eval_rays(ray_origins, ray_directions, intervals)
        bundle = self.volume_renderer(radiance, intervals, ray_directions)
        if self.training:                                         # model_buff.py:66-68
            self.tree.ray_batch_integration(self.global_step, indices[mask], bundle.weights[mask].detach(),
                                            bundle.mask_weights[mask].detach())
        return bundle

    def query(self, ray_batch):
        return self.forward(ray_batch)

    def training_step(self, ray_batch, batch_idx):
        """model_buff.py:79-124 without the TensorBoard loggers: one forward over the whole ray batch, MSE loss,
        tree consolidation on its schedule."""
        bundle = DataBundle.deserialize(ray_batch).to_ray_batch()
        dev = self.model.layer1.weight.device
        out = self.forward((bundle.ray_origins, bundle.ray_directions, bundle.ray_bounds))
        self.check_early_stopping(out.rgb_map)
        loss = self.loss(out.rgb_map, bundle.ray_targets.to(dev))
        log = {"train/loss": loss, "train/psnr": self.criterion_psnr(loss)}
        if self.tree.ticked(self.global_step):
            self.tree.consolidate()
        trainer = getattr(self, "trainer", None)
        try:
            lr = trainer.optimizers[0].param_groups[0]["lr"]
        except (AttributeError, IndexError, TypeError):
            lr = self.cfg.optimizer.lr
        return {"loss": loss, "log": {**log, "train/lr": lr}}

    def validation_step(self, image_ray_batch, batch_idx):
        """model_buff.py:126-164: one whole image in chunks of cfg.nerf.validation.chunksize, loss = sum of the chunk
        MSEs over the FLOAT batch count (the reference's quirk); images go to the logger if one is attached."""
        from ..nerf.nerf_helpers import cast_to_image
        bundle = DataBundle.deserialize(image_ray_batch).to_ray_batch()
        dev = self.model.layer1.weight.device
        batch_size = self.cfg.nerf.validation.chunksize
        batch_count = bundle.ray_targets.shape[0] / batch_size
        loss, rgb_chunks = 0.0, []
        rays = bundle.ray_targets.shape[0]
        per_ray = bundle.ray_origins.shape[0] == rays and rays > 1   # the reference passes origins unsliced (and fails here)
        for i in range(0, rays, batch_size):
            sl = slice(i, i + batch_size)
            origins = bundle.ray_origins[sl] if per_ray else bundle.ray_origins
            out = self.forward((origins, bundle.ray_directions[sl], bundle.ray_bounds))
            loss += self.loss(out.rgb_map, bundle.ray_targets[sl].to(dev))
            rgb_chunks.append(out.rgb_map)
        loss /= batch_count
        rgb_map = torch.cat(rgb_chunks, 0)
        experiment = getattr(getattr(self, "logger", None), "experiment", None)
        if experiment is not None and bundle.hwf is not None:
            experiment.add_image("validation/rgb_coarse/" + str(batch_idx),
                                 cast_to_image(rgb_map.view(bundle.hwf[0], bundle.hwf[1], 3)), self.global_step)
            experiment.add_image("validation/img_target/" + str(batch_idx),
                                 cast_to_image(bundle.ray_targets.view(bundle.hwf[0], bundle.hwf[1], 3)), self.global_step)
        return {"val_loss": loss, "log": {"validation/loss": loss, "validation/psnr": self.criterion_psnr(loss)}}

    def on_save_checkpoint(self, checkpoint):
        checkpoint["tree"] = self.tree.serialize()

    def on_load_checkpoint(self, checkpoint):
        self.tree.deserialize(checkpoint["tree"])
