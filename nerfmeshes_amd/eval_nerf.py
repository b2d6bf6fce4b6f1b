"""Test-set evaluation with the reference's entry points (mirror of /root/reference/src/eval_nerf.py):

* `eval_nerf(model, config_args, cfg, device)` (eval_nerf.py:23-105) -- `BlenderDataset(cfg, TEST)` through a
  `DataLoader(batch_size=1)`, per image chunks of `cfg.nerf.validation.chunksize` rays through `model.query`, the
  per-image loss `sum(mse(chunk)) / (num_rays / chunksize)` (a FLOAT batch count -- the reference's quirk), dataset
  loss = mean over images, PSNR = -10 log10(loss); `--save-images`, `--save-disparity`, `--synthesis-images`.
* the same command line (`--log-checkpoint --checkpoint --save-dir --save-images --save-disparity --synthesis-images`).

Additions: `--views N` evaluates N synthetic orbit views instead of a dataset (`eval_views`; nothing to read from disk,
ray directions generated on the GPU); under `torch.distributed.run` the images are dealt round-robin to the ranks and
the per-image losses all-gathered (`--gpus`-style scaling: launch one process per GPU)."""
import argparse
import os
from pathlib import Path

import torch
import torch.nn.functional as F
from torch.utils.data import DataLoader

from . import hip_ops, models, synthetic
from .data.data_helpers import DataBundle
from .data.datasets import BlenderDataset, DatasetType
from .lightning_modules import PathParser
from .nerf.nerf_helpers import batchify, cast_to_disparity_image, cast_to_pil_image, mse2psnr


def synthetic_views(count, height=800, width=800, focal=synthetic.LEGO_FOCAL_800, with_targets=True):
    for i, pose in enumerate(synthetic.orbit_poses(count)):
        tgt = torch.from_numpy(synthetic.pseudo_targets(height * width, seed=42 + i)) if with_targets else None
        yield pose, height, width, focal, tgt


def render_view(model, pose, height, width, focal, bounds, chunksize, device="cuda"):
    """One view in chunks of `chunksize` rays (eval_nerf.py:62-65).  A NeRFModel in deterministic eval generates the
    rays inside the kernels from the pose (query_view); other models get get_ray_bundle's rays."""
    rgb, disp = [], []
    n = height * width
    in_kernel_rays = hasattr(model, "query_view") and model.can_query_view()   # a failing query_view is an error, not a fallback
    origin = dirs = None
    for s in range(0, n, chunksize):
        count = min(chunksize, n - s)
        out = None
        if in_kernel_rays:
            out = model.query_view(pose, height, width, focal, bounds, first=s, count=count)
        if out is None:
            if dirs is None:
                origin, dirs = hip_ops.ray_bundle(pose, height, width, focal, device=device)
                origin = origin[None]
            out = model.query((origin, dirs[s:s + count], bounds))
        rgb.append(out.rgb_map)
        disp.append(out.disp_map)
    return torch.cat(rgb, 0), torch.cat(disp, 0)


def _imwrite(path, image):
    try:
        import imageio
        imageio.imwrite(path, image)
    except ImportError:
        from PIL import Image
        Image.fromarray(image).save(path)


def eval_nerf(model, config_args, cfg, device):
    """eval_nerf.py:23-105, call for call: dataset -> DataLoader -> `DataBundle.deserialize(...).to_ray_batch()` ->
    `batchify(directions, targets)` -> `model.query((origins, directions_chunk, bounds))` (bounds stay on the host, as
    the reference passes them) -> loss bookkeeping / image files.  Returns the dataset loss (None with
    `--synthesis-images`; the reference returns nothing and prints it)."""
    from . import dist as nd
    rank, world = nd.world()
    dataset = BlenderDataset(cfg, type=DatasetType.TEST)
    if config_args.synthesis_images:
        dataset.synthesis()
    data_loader = DataLoader(dataset, batch_size=1)
    root_dir = Path(config_args.save_dir) / cfg.experiment.id
    images_dir, targets_dir, disparity_dir = root_dir / "images", root_dir / "targets", root_dir / "disparity"
    if config_args.save_images:
        os.makedirs(images_dir, exist_ok=True)
        os.makedirs(targets_dir, exist_ok=True)
    if config_args.save_disparity:
        os.makedirs(disparity_dir, exist_ok=True)
    score = not config_args.synthesis_images
    losses = []
    for img_nr, ray_batch in enumerate(data_loader):
        if img_nr % world != rank:                 # images are independent: round-robin over the ranks
            continue
        bundle = DataBundle.deserialize(ray_batch).to_ray_batch()
        batch_size = cfg.nerf.validation.chunksize
        batch_count = bundle.ray_directions.shape[0] / batch_size          # float (eval_nerf.py:57)
        loss, rgb_map, disp_map = 0, [], []
        origins = bundle.ray_origins.to(device)
        per_ray = origins.shape[0] == bundle.ray_directions.shape[0] and origins.shape[0] > 1      # NDC caches
        chunks = batchify(bundle.ray_directions, bundle.ray_targets, origins if per_ray else None,
                          batch_size=batch_size, device=device, progress=False)
        for ray_directions, ray_targets, ray_origins in chunks:
            out = model.query((ray_origins if per_ray else origins, ray_directions, bundle.ray_bounds))
            rgb_map.append(out.rgb_map)
            disp_map.append(out.disp_map)
            if score:
                loss += F.mse_loss(out.rgb_map, ray_targets)
        if score:
            loss /= batch_count
            losses.append(loss)
        rgb_map, disp_map = torch.cat(rgb_map, 0), torch.cat(disp_map, 0)
        height, width = int(bundle.hwf[0]), int(bundle.hwf[1])
        if config_args.save_images:
            _imwrite(os.path.join(images_dir, f"{img_nr:04d}.png"), cast_to_pil_image(rgb_map.view(height, width, 3)))
            if score:
                _imwrite(os.path.join(targets_dir, f"{img_nr:04d}.png"),
                         cast_to_pil_image(bundle.ray_targets.view(height, width, 3)))
        if config_args.save_disparity:
            _imwrite(os.path.join(disparity_dir, f"{img_nr:04d}.png"),
                     cast_to_disparity_image(disp_map.view(height, width), white_background=True))
        if score:
            print(f"[EVAL] Iter: {img_nr} Loss MSE {loss} / PSNR: {mse2psnr(loss)}")
    if not score:
        return None
    if world > 1:
        counts = nd.round_robin_counts(len(dataset), world)
        mine = torch.stack(losses).to(device).float() if losses else torch.empty(0, dtype=torch.float32, device=device)
        flat = nd.all_gather_rows(mine, counts)
        starts = [sum(counts[:r]) for r in range(world)]
        losses = [flat[starts[i % world] + i // world] for i in range(len(dataset))]     # back to image order
    total_loss = torch.stack(losses).mean()
    print(f"Dataset loss MSE: {total_loss} / PSNR: {mse2psnr(total_loss)}")
    return total_loss


def eval_views(model, views, cfg, device="cuda", chunksize=None, render_chunk=None):
    """The same bookkeeping over an iterable of `(c2w pose, height, width, focal, targets | None)` -- e.g.
    `synthetic_views()` -- with ray directions generated on the GPU.  Returns (per-view losses, dataset loss, dataset
    PSNR, last rgb map).  `targets` may be a callable `(view_nr, rgb) -> (rays, 3) targets` (bench.py photographs the
    render itself).  `render_chunk` renders in larger calls than the bookkeeping's `chunksize` -- the pixels do not depend
    on the chunking (tests/test_gpu_parity.py::test_full_size_view_properties), the loss is still summed per `chunksize`
    rays and divided by the float batch count."""
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
        rgb, _ = render_view(model, pose, h, w, focal, bounds, render_chunk or chunksize, device)
        if callable(targets):
            targets = targets(view_nr, rgb)
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


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--log-checkpoint", type=str, default=None,
                   help="Training log path with the config and checkpoints to load existent configuration.")
    p.add_argument("--checkpoint", type=str, default="model_last.ckpt",
                   help="Load existent configuration from the latest checkpoint by default.")
    p.add_argument("--save-dir", type=str, default=".", help="Save assets to this directory, if specified.")
    p.add_argument("--save-images", action="store_true", default=False, help="Save view images.")
    p.add_argument("--save-disparity", action="store_true", default=False, help="Save disparity images.")
    p.add_argument("--synthesis-images", action="store_true", default=False,
                   help="Synthesis new views 360 degrees around the neural scene.")
    p.add_argument("--views", type=int, default=0,
                   help="(addition) evaluate this many synthetic orbit views instead of the dataset of the config")
    p.add_argument("--chunksize", type=int, default=None, help="(addition) override cfg.nerf.validation.chunksize")
    p.add_argument("--precision", choices=("f32", "bf16x3"), default="f32",
                   help="(addition) arithmetic of the network kernels: f32 = the reference's (default); bf16x3 = fp32 products "
                        "emulated on the bf16 matrix pipe (fp32-class error, |dPSNR| <= 1e-4 dB on the parity fixtures, ~1.8x faster)")
    return p


def main(argv=None):
    args = build_parser().parse_args(argv)
    if not torch.cuda.is_available():
        raise SystemExit("eval_nerf needs a MI355X: the HIP path has no CPU fallback")
    from . import dist as nd
    rank, world, device = nd.init_from_env()          # one process per GPU under torch.distributed.run
    pp = PathParser()
    cfg, _ = pp.parse(None, args.log_checkpoint, None, args.checkpoint)
    if args.chunksize:
        cfg.nerf.validation.chunksize = args.chunksize
    print(f"Loading model from {pp.checkpoint_path}")
    model = getattr(models, cfg.experiment.model).load_from_checkpoint(pp.checkpoint_path).eval().to(device)
    model.set_precision(args.precision)
    try:
        with torch.no_grad():
            if args.views > 0:
                return eval_views(model, synthetic_views(args.views), cfg, device, args.chunksize)
            return eval_nerf(model, args, cfg, device)
    finally:
        nd.shutdown()


if __name__ == "__main__":
    main()
