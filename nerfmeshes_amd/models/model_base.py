"""BaseModel: the reference's Lightning base class surface (/root/reference/src/models/model_base.py:17-187)
over the HIP hot path.  `pytorch_lightning` is used when importable (train_nerf.py wiring); otherwise
`nerfmeshes_amd.lightning_compat.LightningModule` provides `.device`, `.hparams`, `.trainer` and `load_from_checkpoint`
with the same checkpoint layout (`state_dict`, `hyper_parameters`, plus BuFF's `tree`)."""
import torch
from torch.utils.data import DataLoader

from ..data.datasets import BlenderDataset, ColmapDataset, DatasetType
from ..nerf import CfgNode, VolumeRenderer, mse2psnr
from .model_helpers import flatten_dict, nest_dict

try:  # pragma: no cover - not installed offline
    import pytorch_lightning as pl
    _Base = pl.LightningModule
    HAVE_LIGHTNING = not getattr(pl, "__version__", "").endswith("nerfmeshes_amd")
except Exception:  # noqa: BLE001
    from ..lightning_compat import LightningModule as _Base   # inference + checkpoint I/O + the Trainer stand-in's hooks
    HAVE_LIGHTNING = False


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

    def set_precision(self, precision):
        """Arithmetic of the inference kernels of every network of this model: "f32" (default; the reference's fp32) or
        the opt-in "bf16x3" (hip_ops.HipMLP: fp32 products emulated on the bf16 matrix pipe, fp32-class error, ~1.8x the
        throughput; the shipped 64- / 128- / 256-wide shapes; a network without such a kernel says so when it is first used).  Training and the
        mesh grid of `mesh_nerf` always run in fp32."""
        from ..hip_ops import PRECISIONS
        if precision not in PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(PRECISIONS)}, got {precision!r}")
        for m in self.modules():
            if hasattr(m, "hip") and hasattr(m, "precision"):
                m.precision = precision
        return self

    def setup(self, stage):
        """model_base.py:42-59: load both splits, then size the Trainer from the config -- `train_iters` optimizer steps
        (min = max), the matching epoch count, validation every `validate_every` steps expressed in epochs."""
        self.load_train_dataset()
        self.load_val_dataset()
        steps, per_epoch = self.cfg.experiment.train_iters, len(self.train_dataset)
        self.trainer.min_steps = self.trainer.max_steps = steps
        self.trainer.min_epochs = self.trainer.max_epochs = steps // per_epoch
        self.trainer.check_val_every_n_epoch = self.cfg.experiment.validate_every // per_epoch

    def query(self, ray_batch):
        raise NotImplementedError

    def sample_points(self, points, rays=None, **kwargs):
        """model_base.py:65-73: finest network on explicit points (N,3) / view dirs (N,3) -> (N,4)."""
        results = self.get_model().forward(points, rays, **kwargs)
        return results[0] if isinstance(results, tuple) else results

    # ---- data feed (model_base.py:105-148): one sample = one image, so both loaders run with batch_size 1 ----
    def load_dataset(self, dataset_type):
        kind = self.cfg.dataset.type
        if kind == "blender":
            return BlenderDataset(self.cfg, type=dataset_type)
        if kind == "colmap":
            return ColmapDataset(self.cfg, type=dataset_type)
        if kind == "scannet":
            raise NotImplementedError
        return None

    def load_train_dataset(self):
        self.train_dataset = self.load_dataset(DatasetType.TRAIN)

    def load_val_dataset(self):
        self.val_dataset = self.load_dataset(DatasetType.VALIDATION)
        self.val_num_samples = self.cfg.nerf.validation.num_samples
        if self.val_num_samples != -1:
            self.val_num_samples = max(min(len(self.val_dataset), self.val_num_samples), 1)

    def _loader(self, dataset, sampler=None):
        # datasets that generate rays touch the GPU in __getitem__/__init__ only through already-materialised host
        # tensors, so worker processes are safe; the cache files are read by the workers
        return DataLoader(dataset, batch_size=1, shuffle=False, sampler=sampler, pin_memory=False,
                          num_workers=self.cfg.dataset.num_workers)

    def train_dataloader(self):
        return self._loader(self.train_dataset)

    def val_dataloader(self):
        """model_base.py:136-148: `nerf.validation.num_samples` random images (with replacement) per validation run,
        -1 for the whole split."""
        sampler = None
        if self.val_num_samples != -1:
            sampler = torch.utils.data.RandomSampler(self.val_dataset, replacement=True, num_samples=self.val_num_samples)
        return self._loader(self.val_dataset, sampler)

    # ---- training-side hooks (model_base.py:150-187); NeRFModel implements training_step / validation_step ----
    def get_scheduler(self, optimizer):
        gamma, step_size = self.cfg.scheduler.options.gamma, self.cfg.scheduler.options.step_size
        return torch.optim.lr_scheduler.LambdaLR(optimizer, lr_lambda=lambda step: gamma ** (step / step_size))

    def configure_optimizers(self):
        from .. import train_ops
        optimizer = train_ops.make_optimizer(self.cfg.optimizer.type, self.parameters(), self.cfg.optimizer.lr)
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
