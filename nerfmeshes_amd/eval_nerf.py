"""Test-set render loop with the reference's bookkeeping (mirror of /root/reference/src/eval_nerf.py:23-105):
per view, chunks of `cfg.nerf.validation.chunksize` rays through `model.query`, the per-view loss
`sum(mse(chunk)) / (num_rays / chunksize)` (a FLOAT batch count -- the reference's quirk), dataset loss = mean
over views, PSNR = -10 log10(loss).

The reference couples this loop to `BlenderDataset` (file I/O, out of scope); here the views come from any
iterable of `(c2w pose, height, width, focal, targets | None)` -- e.g. `synthetic_views()` -- and ray
directions are generated on the GPU.  `--gpus N` semantics: launch under torch.distributed.run; views are
sharded round-robin over ranks and the per-view losses all-gathered."""
import argparse

import torch

from . import hip_ops, models, synthetic
from .lightning_modules import PathParser
from .nerf.nerf_helpers import mse2psnr


def synthetic_views(count, height=800, width=800, focal=synthetic.LEGO_FOCAL_800, with_targets=True):
    for i, pose in enumerate(synthetic.orbit_poses(count)):
        tgt = torch.from_numpy(synthetic.pseudo_targets(height * width, seed=42 + i)) if with_targets else None
        yield pose, height, width, focal, tgt


def render_view(model, pose, height, width, focal, bounds, chunksize, device="cuda"):
    origin, dirs = hip_ops.ray_bundle(pose, height, width, focal, device=device)
    origin = origin[None]
    rgb, disp = [], []
    for s in range(0, dirs.shape[0], chunksize):
        out = model.query((origin, dirs[s:s + chunksize], bounds))
        rgb.append(out.rgb_map)
        disp.append(out.disp_map)
    return torch.cat(rgb, 0), torch.cat(disp, 0)


def eval_nerf(model, views, cfg, device="cuda", chunksize=None):
    """Returns (per-view losses, dataset loss, dataset PSNR, last rgb map)."""
    chunksize = chunksize or cfg.nerf.validation.chunksize
    bounds = torch.tensor([cfg.dataset.near, cfg.dataset.far], dtype=torch.float32)
    from . import dist as nd
    rank, world = nd.world()
    losses, rgb = [], None
    for view_nr, (pose, h, w, focal, targets) in enumerate(views):
        if view_nr % world != rank:           # views are independent: round-robin over the ranks
            continue
        rgb, _ = render_view(model, pose, h, w, focal, bounds, chunksize, device)
        if targets is not None:
            targets = targets.to(device)
            batch_count = rgb.shape[0] / chunksize              # float: 640000 / 2048 = 312.5 (eval_nerf.py:57)
            loss = 0
            for s in range(0, rgb.shape[0], chunksize):
                loss += torch.nn.functional.mse_loss(rgb[s:s + chunksize], targets[s:s + chunksize])
            loss /= batch_count
            losses.append(loss)
            print(f"[EVAL] Iter: {len(losses) - 1} Loss MSE {loss} / PSNR: {mse2psnr(loss)}")
    if world > 1 and losses:                  # assemble the per-view losses of all ranks (same count per rank assumed)
        mine = torch.stack(losses).to(device)
        losses = list(nd.all_gather_rows(mine, [mine.shape[0]] * world))
    total = torch.stack(losses).mean() if losses else None
    if total is not None:
        print(f"Dataset loss MSE: {total} / PSNR: {mse2psnr(total)}")
    return losses, total, (mse2psnr(total) if total is not None else None), rgb


def main(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--log-checkpoint", type=str, default=None)
    p.add_argument("--checkpoint", type=str, default="model_last.ckpt")
    p.add_argument("--views", type=int, default=4, help="synthetic orbit views to render")
    p.add_argument("--chunksize", type=int, default=None)
    args = p.parse_args(argv)
    if not torch.cuda.is_available():
        raise SystemExit("eval_nerf needs a MI355X: the HIP path has no CPU fallback")
    pp = PathParser()
    cfg, _ = pp.parse(None, args.log_checkpoint, None, args.checkpoint)
    model = getattr(models, cfg.experiment.model).load_from_checkpoint(pp.checkpoint_path).eval().to("cuda")
    with torch.no_grad():
        eval_nerf(model, synthetic_views(args.views), cfg, "cuda", args.chunksize)


if __name__ == "__main__":
    main()
