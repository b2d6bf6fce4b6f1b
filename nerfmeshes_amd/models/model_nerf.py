"""NeRFModel: coarse + fine hierarchical renderer (mirror of /root/reference/src/models/model_nerf.py:10-86)."""
import torch

from .. import hip_ops
from ..nerf import RaySampleInterval, SamplePDF, models as nerf_models
from ..nerf.modules import OutputBundle
from .model_base import BaseModel


def create_models(cfg):
    coarse = getattr(nerf_models, cfg.models.coarse_type)(**cfg.models.coarse)
    fine = None
    if hasattr(cfg.models, "fine") and cfg.models.use_fine:
        fine = getattr(nerf_models, cfg.models.fine_type)(**cfg.models.fine)
    return coarse, fine


class NeRFModel(BaseModel):
    def __init__(self, cfg, *args, **kwargs):
        super().__init__(cfg, *args, **kwargs)
        self.model_coarse, self.model_fine = create_models(self.cfg)
        # sample counts come from the *train* section even in eval (model_nerf.py:30-31)
        self.sample_pdf = SamplePDF(self.cfg.nerf.train.num_fine)
        self.sampler = RaySampleInterval(self.cfg.nerf.train.num_coarse)

    def get_model(self):
        return self.model_fine if self.model_fine is not None else self.model_coarse

    def forward(self, x):
        """x = (ray_origins (1|R,3), ray_directions (R,3), ray_bounds (2,) possibly on the host)
        -> (coarse OutputBundle, fine OutputBundle | None); one nm_render_rays call."""
        ray_origins, ray_directions, bounds = x
        near, far = bounds
        nerf_cfg = self.cfg.nerf.train if self.model_coarse.training else self.cfg.nerf.validation
        if nerf_cfg.perturb or (self.volume_renderer.train_radiance_field_noise_std > 0 and self.training):
            raise NotImplementedError("perturb / radiance noise (training mode) are not implemented on the HIP "
                                      "path; call model.eval() (all shipped validation configs are deterministic)")
        dev = self.model_coarse.layer1.weight.device
        near = torch.as_tensor(near, dtype=torch.float32).reshape(-1)
        far = torch.as_tensor(far, dtype=torch.float32).reshape(-1)
        fine = self.model_fine.hip() if self.model_fine is not None else None
        cb, fb = hip_ops.render_rays(
            self.model_coarse.hip(), fine, ray_origins.to(dev), ray_directions, near, far,
            self.sampler.point_intervals.reshape(-1), self.sample_pdf.u if fine is not None else None,
            lindisp=bool(nerf_cfg.lindisp), white_background=bool(self.volume_renderer.white_background),
            training=bool(self.volume_renderer.training),
            attenuation_threshold=self.volume_renderer.attenuation_threshold)
        return OutputBundle(**cb), (OutputBundle(**fb) if fb is not None else None)

    def query(self, ray_batch):
        coarse, fine = self.forward(ray_batch)
        return fine if fine is not None else coarse

    # names used by BASELINE.json's north_star (upstream krrish94/nerf-pytorch); absent in this reference
    run_iter = forward
    predict_and_render_radiance = forward
