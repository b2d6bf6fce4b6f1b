"""Experiment plumbing with the reference's names (/root/reference/src/lightning_modules.py):
`PathParser` (config / log-dir / checkpoint path resolution) and `LoggerCallback` (progress + metric lines).
Pure host-side glue -- nothing here touches the GPU path."""
import os
import re
from pathlib import Path

import yaml

from .models.model_helpers import nest_dict
from .nerf import CfgNode

try:  # pragma: no cover - not installed offline
    from pytorch_lightning.callbacks import Callback
    from pytorch_lightning.loggers import TensorBoardLogger
    HPARAMS_FILE = TensorBoardLogger.NAME_HPARAMS_FILE
except Exception:  # noqa: BLE001
    from .lightning_compat import Callback, TensorBoardLogger   # same directory layout, scalars to metrics.jsonl
    HPARAMS_FILE = TensorBoardLogger.NAME_HPARAMS_FILE


class LoggerCallback(Callback):
    """Prints `[TRAIN] Iter: n LOSS: ... PSNR: ...` every cfg.experiment.print_every steps, using acronyms of
    the metric names when cfg.logging.use_acronyms (lightning_modules.py:14-143)."""

    def __init__(self, cfg, global_progress=True, leave_global_progress=True):
        super().__init__()
        self.cfg = cfg
        self.global_progress = global_progress
        self.leave_global_progress = leave_global_progress
        self.global_pb, self.val_pb = None, None

    @staticmethod
    def format(value):
        return "{:.6f}".format(value)

    @staticmethod
    def extract_acronym(name):
        tokens = re.split(r"[\s_/]", name)
        return "".join(t[0] for t in tokens[1:]) if len(tokens) > 2 else tokens[-1]

    def get_global_step(self, trainer):
        return trainer.batch_idx + 1 + trainer.current_epoch * len(trainer.train_dataloader)

    def extract_metrics(self, trainer, step=-1, type="train"):
        metrics = {k: v for k, v in trainer.callback_metrics.items() if type in k}
        if step != -1:
            trainer.logger.log_metrics(metrics, step=step)
        if self.cfg.logging.use_acronyms:
            metrics = {self.extract_acronym(k): v for k, v in metrics.items()}
        return " ".join(f"{k.upper()}: {self.format(v)}" for k, v in metrics.items())

    def _trackers(self, trainer, pl_module):
        from tqdm import tqdm
        if self.global_pb is None:
            self.global_pb = tqdm(desc="TRAIN", total=trainer.max_steps, position=0, initial=trainer.global_step,
                                  leave=self.leave_global_progress, disable=not self.global_progress)
        if self.val_pb is None:
            self.val_pb = tqdm(desc="VALID", total=getattr(pl_module, "val_num_samples", None), position=1,
                               leave=self.leave_global_progress, disable=not self.global_progress)

    def on_sanity_check_start(self, trainer, pl_module):
        self._trackers(trainer, pl_module)

    def on_train_epoch_start(self, trainer, pl_module):
        self._trackers(trainer, pl_module)
        self.global_pb.unpause()

    def on_train_batch_end(self, trainer, pl_module, batch, batch_idx, dataloader_idx):
        step = self.get_global_step(trainer)
        if step % self.cfg.experiment.print_every == 0:
            self.global_pb.write(f"[TRAIN] Iter: {step} {self.extract_metrics(trainer, step, 'train')}")
        self.global_pb.update(1)

    def on_validation_start(self, trainer, pl_module):
        self.val_pb.reset()
        self.val_pb.write("  [VAL] =======> Iter: " + str(self.get_global_step(trainer)))

    def on_validation_batch_end(self, trainer, pl_module, batch, batch_idx, dataloader_idx):
        self.val_pb.update(1)

    def on_validation_epoch_end(self, trainer, pl_module):
        self.val_pb.write(self.extract_metrics(trainer, -1, "validation"))

    def on_validation_end(self, trainer, pl_module):
        self.val_pb.clear()

    def on_fit_end(self, trainer, pl_module):
        self.global_pb.close()
        self.val_pb.close()


class PathParser:
    """`parse(config_path | log_path, ...)` -> (cfg, logger).  With a log path
    `<logdir>/<exp>/<run>/version_N` the config is read from its `hparams.yaml` (flat dotted keys) and
    `checkpoint_path` points at `checkpoints/<checkpoint_name>` (lightning_modules.py:146-222)."""

    LOG_RUN_NAME = "default"
    CHECKPOINT_NAME_LAST = "model_last.ckpt"

    def __init__(self):
        self.root_path = self.config_path = None
        self.log_root_dir = self.log_dir = None
        self.exp_name = self.log_name = self.log_version = None
        self.checkpoint_dir = self.checkpoint_path = None

    def parse(self, config_path=None, log_path=None, run_name=LOG_RUN_NAME, checkpoint_name=CHECKPOINT_NAME_LAST,
              create_logger=False):
        assert (config_path is not None) != (log_path is not None), \
            "Either config or log with checkpoints must be provided, append option --help for more information."
        if log_path is not None:
            self.exp_name, self.log_name, self.log_version = os.path.normpath(log_path).split(os.path.sep)[-3:]
            self.log_dir = Path(log_path)
            self.config_path = str(self.log_dir / HPARAMS_FILE)
        else:
            self.config_path = config_path
        with open(self.config_path, "r") as fh:
            cfg = CfgNode(nest_dict(yaml.safe_load(fh), sep="."))
        self.root_path = Path(cfg.experiment.logdir)
        if log_path is None:
            self.exp_name, self.log_name = cfg.experiment.id, run_name
        self.log_root_dir = str(self.root_path / self.exp_name)
        logger = None
        if create_logger:
            os.makedirs(Path(self.log_root_dir) / self.log_name, exist_ok=True)
            logger = TensorBoardLogger(self.log_root_dir, self.log_name, version=self.log_version)
            if hasattr(logger, "resolve_version"):      # the stand-in: one broadcast, here, where every rank passes
                from . import dist as nd
                nd.init_from_env()                       # joins the launcher's process group (no-op without one)
                logger.resolve_version()
            self.log_dir = Path(logger.log_dir)
        if self.log_dir is None:   # config-only parse without a logger: mirror Lightning's first version dir
            self.log_dir = Path(self.log_root_dir) / self.log_name / "version_0"
        print(f"Current log dir {self.log_dir}")
        self.checkpoint_dir = self.log_dir / "checkpoints"
        os.makedirs(self.checkpoint_dir, exist_ok=True)
        if log_path is not None:
            self.checkpoint_path = str(self.checkpoint_dir / checkpoint_name)
        return cfg, logger
