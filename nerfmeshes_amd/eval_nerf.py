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
    """One view in chunks of `chunksize` rays (eval_nerf.py:62-65).  A NeRFModel in deterministic eval generates the
    rays inside the kernels from the pose (query_view); other models get get_ray_bundle's rays."""
    rgb, disp = [], []
    n = height * width
    try_view = hasattr(model, "query_view")
    origin = dirs = None
    for s in range(0, n, chunksize):
        count = min(chunksize, n - s)
        out = None
        if try_view:
            try:
                out = model.query_view(pose, height, width, focal, bounds, first=s, count=count)
            except RuntimeError:
                try_view = False
        if out is None:
            if dirs is None:
                origin, dirs = hip_ops.ray_bundle(pose, height, width, focal, device=device)
                origin = origin[None]
            out = model.query((origin, dirs[s:s + count], bounds))
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
    scored = [0] * world                      # views WITH targets per rank (every rank walks the whole iterable)
    slots = []                                # (owner rank, index among that rank's scored views), in view order
    for view_nr, (pose, h, w, focal, targets) in enumerate(views):
        owner = view_nr % world               # views are independent: round-robin over the ranks
        if targets is not None:
            slots.append((owner, scored[owner]))
            scored[owner] += 1
        if owner != rank:
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
    if world > 1:
        # Ranks hold DIFFERENT numbers of views when the view count is not a multiple of the world size (possibly
        # none): every rank enters the one collective with its (scored[rank],) tensor -- empty included -- and the
        # rank-major result is re-interleaved into view order.
        mine = torch.stack(losses).to(device) if losses else torch.empty(0, dtype=torch.float32, device=device)
        flat = nd.all_gather_rows(mine.to(torch.float32), scored)
        starts = [sum(scored[:r]) for r in range(world)]
        losses = [flat[starts[r] + k] for r, k in slots]
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
    from . import dist as nd
    rank, world, device = nd.init_from_env()          # one process per GPU under torch.distributed.run
    pp = PathParser()
    cfg, _ = pp.parse(None, args.log_checkpoint, None, args.checkpoint)
    model = getattr(models, cfg.experiment.model).load_from_checkpoint(pp.checkpoint_path).eval().to(device)
    try:
        with torch.no_grad():
            return eval_nerf(model, synthetic_views(args.views), cfg, device, args.chunksize)
    finally:
        nd.shutdown()


if __name__ == "__main__":
    main()
