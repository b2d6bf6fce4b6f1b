"""BaseModel: the reference's Lightning base class surface (/root/reference/src/models/model_base.py:17-187)
over the HIP hot path.  `pytorch_lightning` is used when importable (train_nerf.py wiring); otherwise a thin
torch.nn.Module stand-in provides `.device`, `.hparams` and `load_from_checkpoint` with the same checkpoint
layout (`state_dict`, `hyper_parameters`, plus BuFF's `tree`)."""
import os

import torch
import yaml

from ..nerf import CfgNode, VolumeRenderer, mse2psnr
from .model_helpers import flatten_dict, nest_dict

try:  # pragma: no cover - not installed offline
    import pytorch_lightning as pl
    _Base = pl.LightningModule
    HAVE_LIGHTNING = True
except Exception:  # noqa: BLE001
    HAVE_LIGHTNING = False

    class _Base(torch.nn.Module):
        """Minimal LightningModule stand-in (inference + checkpoint I/O only)."""

        def __init__(self, *args, **kwargs):
            super().__init__()
            self.global_step = 0

        @property
        def device(self):
            for t in list(self.parameters()) + list(self.buffers()):
                return t.device
            return torch.device("cpu")

        def on_save_checkpoint(self, checkpoint):
            pass

        def on_load_checkpoint(self, checkpoint):
            pass

        def save_checkpoint(self, path):
            ckpt = {"state_dict": self.state_dict(), "hyper_parameters": dict(self.hparams), "epoch": 0,
                    "global_step": self.global_step}
            self.on_save_checkpoint(ckpt)
            os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
            torch.save(ckpt, path)

        @classmethod
        def load_from_checkpoint(cls, checkpoint_path, map_location=None, **kwargs):
            ckpt = torch.load(checkpoint_path, map_location=map_location or "cpu", weights_only=False)
            hparams = ckpt.get("hyper_parameters")
            if not hparams:  # Lightning also writes <version>/hparams.yaml next to checkpoints/
                side = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(checkpoint_path))), "hparams.yaml")
                with open(side) as fh:
                    hparams = yaml.safe_load(fh)
            model = cls(dict(hparams), **kwargs)
            model.load_state_dict(ckpt["state_dict"])
            model.on_load_checkpoint(ckpt)
            return model


def _with_defaults(flat):
    """Schema drift tolerated by the reference's shipped hparams: older files carry
    `dataset.no_ndc` instead of `dataset.use_ndc` and lack the early-stopping keys."""
    flat = dict(flat)
    if "dataset.use_ndc" not in flat:
        flat["dataset.use_ndc"] = not flat.get("dataset.no_ndc", True)
    flat.setdefault("experiment.use_early_stopping", False)
    flat.setdefault("experiment.early_stopping_step", 25)
    return flat


class BaseModel(_Base):
    def __init__(self, cfg, *args, **kwargs):
        super().__init__(*args, **kwargs)
        flat = _with_defaults(flatten_dict(cfg, sep="."))
        self.cfg = CfgNode(nest_dict(flat, sep="."))
        if HAVE_LIGHTNING:  # pragma: no cover
            self.save_hyperparameters(flat)
        else:
            self.hparams = flat
        self.loss = torch.nn.MSELoss()
        self.criterion_psnr = mse2psnr
        self.volume_renderer = VolumeRenderer(
            self.cfg.nerf.train.radiance_field_noise_std, self.cfg.nerf.validation.radiance_field_noise_std,
            self.cfg.dataset.white_background, attenuation_threshold=1e-5)
        self.train_dataset, self.val_dataset = None, None

    def get_model(self):
        raise NotImplementedError

    def query(self, ray_batch):
        raise NotImplementedError

    def sample_points(self, points, rays=None, **kwargs):
        """model_base.py:65-73: finest network on explicit points (N,3) / view dirs (N,3) -> (N,4)."""
        results = self.get_model().forward(points, rays, **kwargs)
        return results[0] if isinstance(results, tuple) else results

    # ---- training-side hooks (model_base.py:150-187); NeRFModel implements training_step / validation_step ----
    def get_scheduler(self, optimizer):
        gamma, step_size = self.cfg.scheduler.options.gamma, self.cfg.scheduler.options.step_size
        return torch.optim.lr_scheduler.LambdaLR(optimizer, lr_lambda=lambda step: gamma ** (step / step_size))

    def configure_optimizers(self):
        optimizer = getattr(torch.optim, self.cfg.optimizer.type)(self.parameters(), lr=self.cfg.optimizer.lr)
        if hasattr(torch.optim.lr_scheduler, self.cfg.scheduler.type):
            scheduler = getattr(torch.optim.lr_scheduler, self.cfg.scheduler.type)(optimizer, **self.cfg.scheduler.options)
        else:
            scheduler = self.get_scheduler(optimizer)
        return [optimizer], [{"scheduler": scheduler, "interval": "step", "frequency": 1}]

    def training_step(self, ray_batch, batch_idx):
        raise NotImplementedError

    def validation_epoch_end(self, outputs):
        """model_base.py:75-103 without the pytorch3d chamfer branch: the mean of every logged value."""
        log = {k: torch.stack([torch.as_tensor(o["log"][k]) for o in outputs]).mean() for k in outputs[0]["log"]}
        return {"log": log, "val_loss": torch.stack([torch.as_tensor(o["val_loss"]) for o in outputs]).mean()}

    def check_early_stopping(self, rgb):
        exp = self.cfg.experiment
        if exp.use_early_stopping and self.global_step == exp.early_stopping_step and rgb.sum() < 1e-12:
            print("Model is stuck in local minima, restart the training; exiting now...")
            raise SystemExit(-1)
