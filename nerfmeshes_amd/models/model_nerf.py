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
        dev = self.model_coarse.layer1.weight.device
        near = torch.as_tensor(near, dtype=torch.float32).reshape(-1)
        far = torch.as_tensor(far, dtype=torch.float32).reshape(-1)
        vr = self.volume_renderer
        noise_std = vr.train_radiance_field_noise_std if vr.training else vr.val_radiance_field_noise_std
        if nerf_cfg.perturb or noise_std > 0.0 or self.model_coarse.needs_grad():
            return self._forward_stochastic(ray_origins.to(dev), ray_directions, near, far, nerf_cfg, noise_std)
        fine = self.model_fine.hip() if self.model_fine is not None else None
        cb, fb = hip_ops.render_rays(
            self.model_coarse.hip(), fine, ray_origins.to(dev), ray_directions, near, far,
            self.sampler.point_intervals.reshape(-1), self.sample_pdf.u if fine is not None else None,
            lindisp=bool(nerf_cfg.lindisp), white_background=bool(self.volume_renderer.white_background),
            training=bool(self.volume_renderer.training),
            attenuation_threshold=self.volume_renderer.attenuation_threshold)
        return OutputBundle(**cb), (OutputBundle(**fb) if fb is not None else None)

    def _forward_stochastic(self, origins, dirs, near, far, nerf_cfg, noise_std):
        """The same chain stage by stage, differentiable in the networks' parameters, with the training-mode
        randomness of RaySampleInterval (modules.py:171-184), VolumeRenderer (:82-91) and SamplePDF (:224-228);
        the draws are torch's (device generator), the arithmetic is the HIP kernels'."""
        from .. import train_ops
        vr = self.volume_renderer
        dev = self.model_coarse.layer1.weight.device
        dirs = hip_ops._dev32(dirs, dev, "ray_directions").reshape(-1, 3)
        origins = hip_ops._dev32(origins, dev, "ray_origins").reshape(-1, 3)
        rays = dirs.shape[0]
        t = hip_ops.coarse_intervals(self.sampler.point_intervals.reshape(-1).to(dev), near, far, rays,
                                     lindisp=bool(nerf_cfg.lindisp))
        if nerf_cfg.perturb:
            t = train_ops.perturb_intervals(t, torch.rand(t.shape, dtype=t.dtype, device=dev))

        def render(net, t):
            radiance = train_ops.mlp_rays(net, origins, dirs, t)
            noise = torch.randn(t.shape, dtype=t.dtype, device=dev) * noise_std if noise_std > 0.0 else None
            b = train_ops.composite(radiance, t, dirs, noise, vr.attenuation_threshold, bool(vr.white_background))
            if not vr.training:                                         # modules.py:108-109
                b["depth_map"] = torch.where(b["acc_map"] < 1.0, torch.zeros_like(b["depth_map"]), b["depth_map"])
            return OutputBundle(**b)

        coarse = render(self.model_coarse, t)
        if self.model_fine is None:
            return coarse, None
        w = coarse.weights.detach()
        if nerf_cfg.perturb:                                            # det = (perturb == 0.0), modules.py:201
            u = torch.rand(rays, self.sample_pdf.num_samples, dtype=t.dtype, device=dev)
            t_fine = train_ops.sample_pdf_rand(t, w, u)
        else:
            t_fine = hip_ops.sample_pdf(t, w, self.sample_pdf.u)
        return coarse, render(self.model_fine, t_fine)

    def query(self, ray_batch):
        coarse, fine = self.forward(ray_batch)
        return fine if fine is not None else coarse

    # names used by BASELINE.json's north_star (upstream krrish94/nerf-pytorch); absent in this reference
    run_iter = forward
    predict_and_render_radiance = forward
