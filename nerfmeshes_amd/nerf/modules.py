"""Render primitives with the reference's class names and signatures
(/root/reference/src/nerf/modules.py:8-248), dispatching to the gfx950 kernels through the C ABI.

Every forward requires GPU tensors: there is no CPU or eager fallback.  In training mode (reference rows (f)-2) the
stratified jitter, the density noise and the random `u` of the fine resampling are drawn from torch's generator and
handed to the HIP kernels (`train_ops`), and the modules are differentiable through the hand-written backward."""
from dataclasses import dataclass

import torch

from .. import hip_ops


@dataclass
class OutputBundle:
    rgb_map: torch.Tensor = None
    depth_map: torch.Tensor = None
    weights: torch.Tensor = None
    mask_weights: torch.Tensor = None
    acc_map: torch.Tensor = None
    disp_map: torch.Tensor = None


class PositionalEncoding(torch.nn.Module):
    """modules.py:8-37.  Inside FlexibleNeRFModel the encoding is computed in the fused MLP kernel and never
    materialised; called on its own, `forward` runs nm_positional_encoding and returns the same rows the reference
    does ([x | sin(x_c * f_k), coordinate-major | cos(...)])."""

    def __init__(self, num_encoding_functions=6, include_input=True, log_sampling=True):
        super().__init__()
        self.num_encoding_functions = num_encoding_functions
        self.include_input = include_input
        n = num_encoding_functions
        bands = 2.0 ** torch.linspace(0.0, n - 1, n) if log_sampling else torch.linspace(1.0, 2.0 ** (n - 1), n)
        self.register_buffer("frequency_bands", bands)

    def output_size(self):
        return 6 * self.num_encoding_functions + (3 if self.include_input else 0)

    def forward(self, x):
        return hip_ops.positional_encoding(x, self.frequency_bands, self.include_input)


class VolumeRenderer(torch.nn.Module):
    """modules.py:50-121 -> nm_composite (no noise, no autograd) / train_ops.composite (noise, differentiable)."""

    def __init__(self, train_radiance_field_noise_std=0.0, val_radiance_field_noise_std=0.0, white_background=False,
                 attenuation_threshold=1e-3):
        super().__init__()
        self.train_radiance_field_noise_std = train_radiance_field_noise_std
        self.val_radiance_field_noise_std = val_radiance_field_noise_std
        self.attenuation_threshold = attenuation_threshold
        self.white_background = white_background
        self.register_buffer("one_e_10", torch.tensor([1e10]))

    def forward(self, radiance_field, depth_values, ray_directions):
        std = self.train_radiance_field_noise_std if self.training else self.val_radiance_field_noise_std
        if std > 0.0 or (torch.is_grad_enabled() and radiance_field.requires_grad):
            from .. import train_ops
            sigma = radiance_field[..., 3]
            noise = torch.randn(sigma.shape, dtype=sigma.dtype, device=sigma.device) * std if std > 0.0 else None
            out = train_ops.composite(radiance_field, depth_values, ray_directions, noise, self.attenuation_threshold,
                                      self.white_background)
            if not self.training:                                    # modules.py:108-109
                out["depth_map"] = torch.where(out["acc_map"] < 1.0, torch.zeros_like(out["depth_map"]), out["depth_map"])
            return OutputBundle(**out)
        out = hip_ops.composite(radiance_field, depth_values, ray_directions, self.attenuation_threshold,
                                self.white_background, self.training)
        return OutputBundle(**out)


class RaySampleInterval(torch.nn.Module):
    """modules.py:148-186 -> nm_coarse_intervals."""

    def __init__(self, count):
        super().__init__()
        self.count = count
        self.register_buffer("point_intervals", torch.linspace(0.0, 1.0, count)[None, :], persistent=False)

    def forward(self, cfg, ray_count, near, far):
        t = hip_ops.coarse_intervals(self.point_intervals.reshape(-1), near, far, ray_count, bool(cfg.lindisp))
        if cfg.perturb:                                              # modules.py:171-184, torch's random draw
            from .. import train_ops
            t = train_ops.perturb_intervals(t, torch.rand(t.shape, dtype=t.dtype, device=t.device))
        return t


class SamplePDF(torch.nn.Module):
    """modules.py:189-248 -> nm_sample_pdf."""

    def __init__(self, num_samples):
        super().__init__()
        self.num_samples = num_samples
        self.register_buffer("u", torch.linspace(0.0, 1.0, steps=num_samples))

    def forward(self, point_interval, weights, perturb):
        if perturb != 0.0:                                           # det = (perturb == 0.0), modules.py:201
            from .. import train_ops
            u = torch.rand(point_interval.shape[0], self.num_samples, dtype=point_interval.dtype,
                           device=point_interval.device)
            return train_ops.sample_pdf_rand(point_interval, weights.detach(), u)
        return hip_ops.sample_pdf(point_interval, weights, self.u)
