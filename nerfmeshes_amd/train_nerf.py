"""Training entry point with the reference's command line (mirror of /root/reference/src/train_nerf.py:14-108):
`--config <yml>` for a new experiment or `--log-checkpoint <logdir>/<exp>/<run>/version_N` (+ `--checkpoint`) to resume,
`--run-name`, `--gpus`, `--precision`, `--deterministic`, `--use-profiler`.

`PathParser` resolves the directories and creates the logger, the model class comes from `cfg.experiment.model`, a
`ModelCheckpoint(save_top_k=3, save_last=True, monitor="val_loss", prefix="model_")` writes
`<version dir>/checkpoints/model_last.ckpt`, `LoggerCallback` prints progress, and `Trainer.fit(model)` runs
`BaseModel.setup` -> data loaders -> `training_step` (HIP forward + backward) -> optimizer.  `pytorch_lightning`'s
Trainer is used when the package is installed, otherwise `nerfmeshes_amd.lightning_compat` (the same hook order).

Multi-GPU: launch one process per GPU (`python -m torch.distributed.run --nproc-per-node N -m nerfmeshes_amd.train_nerf
...`); `--gpus` is accepted for command-line compatibility and must match the launcher's world size.  A dataset-free
demo of the same training arithmetic is `nerfmeshes_amd.train_synthetic`.
"""
import argparse
import os

import torch

from . import models
from .lightning_modules import LoggerCallback, PathParser

try:  # pragma: no cover - not installed offline
    from pytorch_lightning import Trainer, seed_everything
    from pytorch_lightning.callbacks import ModelCheckpoint
except Exception:  # noqa: BLE001
    from .lightning_compat import ModelCheckpoint, Trainer, seed_everything


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--config", type=str, default=None, help="Path to (.yml) config file if running new experiment.")
    p.add_argument("--log-checkpoint", type=str, default=None,
                   help="Training log path with the config and checkpoints to resume the experiment.")
    p.add_argument("--checkpoint", type=str, default="model_last.ckpt",
                   help="Resume training from the latest checkpoint by default.")
    p.add_argument("--run-name", type=str, default="default", help="Name of the training log run")
    p.add_argument("--gpus", type=int, default=1, help="Amount of Gpus that should be used (one process per GPU)")
    p.add_argument("--precision", type=int, default=32, help="Full precision (32) only on this path.")
    p.add_argument("--deterministic", action="store_true", default=False,
                   help="Run deterministic training, useful for experimenting")
    p.add_argument("--use-profiler", action="store_true", default=False, help="Accepted; profile with rocprofv3 instead")
    return p


def main(argv=None):
    torch.set_printoptions(threshold=100, edgeitems=50, precision=8, sci_mode=False)
    args = build_parser().parse_args(argv)
    if not torch.cuda.is_available():
        raise SystemExit("train_nerf needs a MI355X: the HIP path has no CPU fallback")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} processes")
    path_parser = PathParser()
    cfg, logger = path_parser.parse(args.config, args.log_checkpoint, args.run_name, args.checkpoint, create_logger=True)
    if args.deterministic:
        seed_everything(cfg.experiment.randomseed)
    model = getattr(models, cfg.experiment.model)(cfg)
    checkpoint_callback = ModelCheckpoint(filepath=path_parser.checkpoint_dir, save_top_k=3, save_last=True, verbose=True,
                                          monitor="val_loss", mode="min", prefix="model_")
    trainer = Trainer(
        weights_summary=None, resume_from_checkpoint=path_parser.checkpoint_path, gpus=args.gpus,
        default_root_dir=path_parser.log_dir, logger=logger, num_sanity_val_steps=0,
        checkpoint_callback=checkpoint_callback, row_log_interval=1, log_gpu_memory=None, precision=args.precision,
        profiler=None, fast_dev_run=False, deterministic=args.deterministic, progress_bar_refresh_rate=0,
        accumulate_grad_batches=1, callbacks=[LoggerCallback(cfg)])
    if args.log_checkpoint is not None:
        logger.experiment.add_text("description", cfg.experiment.description, 0)
        logger.experiment.add_text("config", f"\t{cfg.dump()}".replace("\n", "\n\t"), 0)
    trainer.fit(model)
    print("Done!")
    return trainer, model, path_parser


if __name__ == "__main__":
    print(os.getcwd())
    main()
