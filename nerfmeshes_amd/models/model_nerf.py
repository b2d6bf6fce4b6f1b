"""NeRFModel: coarse + fine hierarchical renderer (mirror of /root/reference/src/models/model_nerf.py:10-86)."""
import torch

from .. import hip_ops
from ..nerf import RaySampleInterval, SamplePDF, models as nerf_models
from ..data import DataBundle
from ..nerf.modules import OutputBundle
from ..nerf.nerf_helpers import cast_to_image
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
            # the tape (L*n*H fp32, ~3 GB at 2048 x 192 samples) is recorded only when a backward pass can follow;
            # eval with perturb / validation noise under torch.no_grad() runs the plain inference kernel
            if net.needs_grad():
                radiance = train_ops.mlp_rays(net, origins, dirs, t)
            else:
                radiance = net.hip().eval_rays(origins, dirs, t)
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

    def can_query_view(self):
        """True when `query_view` applies: deterministic inference (no stratified jitter, no density noise, no autograd)."""
        nerf_cfg = self.cfg.nerf.train if self.model_coarse.training else self.cfg.nerf.validation
        vr = self.volume_renderer
        noise_std = vr.train_radiance_field_noise_std if vr.training else vr.val_radiance_field_noise_std
        return not (nerf_cfg.perturb or noise_std > 0.0 or self.model_coarse.needs_grad())

    def query_view(self, pose, height, width, focal, bounds, first=0, count=None):
        """`query` for pixels [first, first+count) of a camera view, the rays generated inside the kernels from the
        pose (nm_render_view; get_ray_bundle + cfg.dataset.use_ndc's ndc_rays): no ray buffers, no per-chunk H2D
        (nerf_helpers.py:128).  Bit-identical to query() on get_ray_bundle's rays.  Deterministic eval only."""
        nerf_cfg = self.cfg.nerf.train if self.model_coarse.training else self.cfg.nerf.validation
        vr = self.volume_renderer
        noise_std = vr.train_radiance_field_noise_std if vr.training else vr.val_radiance_field_noise_std
        if nerf_cfg.perturb or noise_std > 0.0 or self.model_coarse.needs_grad():
            raise RuntimeError("query_view is the deterministic inference path (no perturb / noise / autograd)")
        near, far = bounds
        view = hip_ops.make_view(pose, height, width, focal, ndc_near=1.0 if self.cfg.dataset.use_ndc else None)
        fine = self.model_fine.hip() if self.model_fine is not None else None
        cb, fb = hip_ops.render_view(
            self.model_coarse.hip(), fine, view, torch.as_tensor(near, dtype=torch.float32).reshape(-1),
            torch.as_tensor(far, dtype=torch.float32).reshape(-1), self.sampler.point_intervals.reshape(-1),
            self.sample_pdf.u if fine is not None else None, first=first, count=count, lindisp=bool(nerf_cfg.lindisp),
            white_background=bool(vr.white_background), training=bool(vr.training),
            attenuation_threshold=vr.attenuation_threshold)
        return OutputBundle(**(fb if fb is not None else cb))

    def _chunks(self, bundle, chunk):
        """The reference's manual batching (model_nerf.py:93-112): slices of `chunk` rays; origins are shared
        unless the rays are in NDC.  Tensors already in GPU memory stay there (no host round trip per chunk)."""
        dev = self.device
        origins, dirs, bounds, targets = bundle.ray_origins, bundle.ray_directions, bundle.ray_bounds, bundle.ray_targets
        for i in range(0, targets.shape[0], chunk):
            sl = slice(i, i + chunk)   # moved per chunk, as the reference: a whole image bundle may not fit next to a tape
            yield ((origins[sl] if self.cfg.dataset.use_ndc else origins).to(dev), dirs[sl].to(dev), bounds), targets[sl].to(dev)

    def _current_lr(self):
        trainer = getattr(self, "trainer", None)
        try:
            return trainer.optimizers[0].param_groups[0]["lr"]
        except (AttributeError, IndexError, TypeError):
            return self.cfg.optimizer.lr

    def training_step(self, ray_batch, batch_idx):
        """model_nerf.py:88-151: loss = MSE(coarse) + MSE(fine) averaged over the chunks with the reference's
        float `batch_count`; the returned loss carries the autograd graph of the HIP forward (train_ops)."""
        bundle = DataBundle.deserialize(ray_batch).to_ray_batch()
        chunk = self.cfg.nerf.train.chunksize
        batch_count = bundle.ray_targets.shape[0] / chunk
        coarse_loss, fine_loss = 0, 0
        for rays, target in self._chunks(bundle, chunk):
            coarse, fine = self.forward(rays)
            coarse_loss += self.loss(coarse.rgb_map, target)
            self.check_early_stopping(coarse.rgb_map)
            if self.model_fine is not None:
                fine_loss += self.loss(fine.rgb_map, target)
                self.check_early_stopping(fine.rgb_map)
        coarse_loss /= batch_count
        log = {"train/coarse_loss": coarse_loss, "train/coarse_psnr": self.criterion_psnr(coarse_loss)}
        loss = coarse_loss
        if self.cfg.models.use_fine:
            fine_loss /= batch_count
            # in place, as the reference (model_nerf.py:137): `loss` aliases `coarse_loss`, so the logged
            # train/coarse_loss becomes the total loss while train/coarse_psnr keeps the coarse value
            loss += fine_loss
            log.update({"train/fine_loss": fine_loss, "train/fine_psnr": self.criterion_psnr(fine_loss)})
        return {"loss": loss, "log": {"train/loss": loss, **log, "train/lr": self._current_lr()}}

    def validation_step(self, image_ray_batch, batch_idx):
        """model_nerf.py:153-222: one whole image in validation chunks; images go to the logger if one is attached."""
        bundle = DataBundle.deserialize(image_ray_batch).to_ray_batch()
        chunk = self.cfg.nerf.validation.chunksize
        batch_count = bundle.ray_targets.shape[0] / chunk
        coarse_loss, fine_loss, rgb_c, rgb_f = 0, 0, [], []
        for rays, target in self._chunks(bundle, chunk):
            coarse, fine = self.forward(rays)
            coarse_loss += self.loss(coarse.rgb_map, target)
            rgb_c.append(coarse.rgb_map)
            if self.model_fine is not None:
                fine_loss += self.loss(fine.rgb_map, target)
                rgb_f.append(fine.rgb_map)
        coarse_loss /= batch_count
        loss = coarse_loss
        log = {"validation/coarse_loss": coarse_loss, "validation/coarse_psnr": self.criterion_psnr(coarse_loss)}
        images = {"validation/rgb_coarse/": torch.cat(rgb_c, 0)}
        if self.model_fine is not None:
            fine_loss /= batch_count
            loss += fine_loss                      # in place: the same aliasing as in training_step (model_nerf.py:195)
            log.update({"validation/fine_loss": fine_loss, "validation/fine_psnr": self.criterion_psnr(fine_loss)})
            images["validation/rgb_fine/"] = torch.cat(rgb_f, 0)
        experiment = getattr(getattr(self, "logger", None), "experiment", None)
        if experiment is not None and bundle.hwf is not None:
            for tag, rgb in images.items():
                experiment.add_image(tag + str(batch_idx), cast_to_image(rgb.view(bundle.hwf[0], bundle.hwf[1], 3)),
                                     self.global_step)
            if bundle.ray_targets is not None:
                experiment.add_image("validation/img_target/" + str(batch_idx),
                                     cast_to_image(bundle.ray_targets.view(bundle.hwf[0], bundle.hwf[1], 3)),
                                     self.global_step)
        return {"val_loss": loss, "log": {"validation/loss": loss, **log}}

    # names used by BASELINE.json's north_star (upstream krrish94/nerf-pytorch); absent in this reference
    run_iter = forward
    predict_and_render_radiance = forward
