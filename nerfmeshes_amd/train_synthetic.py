"""Dataset-free training demo over the HIP path: `configure_optimizers()` -> repeated `training_step()` ->
`loss.backward()` -> optimizer / scheduler step, one process per GPU with the gradient all-reduce of
`nerfmeshes_amd.dist` when launched under `torch.distributed.run`.  (The mirror of the reference's command line, with
datasets, the Trainer and checkpoints, is `nerfmeshes_amd.train_nerf`.)

There is no dataset in this environment, so the ray batches come from a *teacher*: the seeded smooth scene
(`synthetic.make_scene_weights`) rendered through the inference path from orbit poses gives the target pixels;
the student is a fresh NeRFModel of the given shape.  `write_blender_scene` stores the same teacher views in the
NeRF-synthetic file layout (transforms_*.json + PNGs) so that the dataset-driven scripts have something to read.

    python -m nerfmeshes_amd.train_synthetic --iters 300 --views 8 --size 100
"""
import argparse
import json
import time

import torch

from . import dist as nd, hip_ops, synthetic as S
from .data import DataBundle, batch_random_sampling
from .nerf import CfgNode, mse2psnr


def teacher_views(num_views, size, device, chunk=65536):
    """Target images of the seeded scene: list of dicts {ray_origins (3,), ray_directions (H,W,3), ray_targets (H,W,3)}."""
    full = dict(num_layers=8, hidden_size=256, skip_step=4, num_encoding_fn_xyz=10, num_encoding_fn_dir=4)
    mlp = hip_ops.HipMLP(S.make_scene_weights(**full), full, device)
    focal = 1111.1111 * size / 800.0
    uc, uf = torch.linspace(0, 1, 64), torch.linspace(0, 1, 128)
    near, far = torch.tensor([2.0]), torch.tensor([6.0])
    views = []
    for pose in S.orbit_poses(num_views):
        origin, dirs = hip_ops.ray_bundle(pose, size, size, focal, 0, size * size, device)
        rgb = []
        for s in range(0, size * size, chunk):
            _, fine = hip_ops.render_rays(mlp, mlp, origin[None], dirs[s:s + chunk], near, far, uc, uf)
            rgb.append(fine["rgb_map"].clone())
        views.append(dict(ray_origins=origin, ray_directions=dirs.view(size, size, 3),
                          ray_targets=torch.cat(rgb).view(size, size, 3), hwf=(size, size, focal)))
    return views


def write_blender_scene(basedir, size=64, counts=(6, 2, 2), device="cuda", views=None):
    """Teacher renders as a NeRF-synthetic scene: `<basedir>/transforms_{train,val,test}.json` + `<split>/r_<i>.png`
    (8-bit RGBA, alpha 255).  Returns {split: list of float (H,W,3) target tensors AS STORED (quantised to 8 bits)}."""
    import json as _json
    import math
    import os
    import numpy as np
    from PIL import Image
    total = sum(counts)
    views = views if views is not None else teacher_views(total, size, device)
    poses = S.orbit_poses(total)
    focal = 1111.1111 * size / 800.0
    stored, k = {}, 0
    for split, n in zip(("train", "val", "test"), counts):
        os.makedirs(os.path.join(basedir, split), exist_ok=True)
        frames, stored[split] = [], []
        for i in range(n):
            rgb8 = (views[k]["ray_targets"].clamp(0, 1) * 255).round().to(torch.uint8).cpu().numpy()
            rgba = np.concatenate([rgb8, np.full(rgb8.shape[:2] + (1,), 255, np.uint8)], -1)
            Image.fromarray(rgba, "RGBA").save(os.path.join(basedir, split, f"r_{i}.png"))
            pose = np.eye(4, dtype=np.float64)
            pose[:3, :4] = np.asarray(poses[k], dtype=np.float64)[:3, :4]
            frames.append({"file_path": f"./{split}/r_{i}", "transform_matrix": pose.tolist()})
            stored[split].append(torch.from_numpy((rgb8 / 255.0).astype(np.float32)))
            k += 1
        with open(os.path.join(basedir, f"transforms_{split}.json"), "w") as fh:
            _json.dump({"camera_angle_x": 2.0 * math.atan(0.5 * size / focal), "frames": frames}, fh)
    return stored


def random_ray_batch(cfg, view, coords):
    dirs, targets = batch_random_sampling(cfg, coords, (view["ray_directions"], view["ray_targets"]))
    bundle = DataBundle(ray_origins=view["ray_origins"], ray_directions=dirs, ray_targets=targets,
                        ray_bounds=torch.tensor([cfg.dataset.near, cfg.dataset.far]))
    return bundle.serialize(["ray_origins", "ray_directions", "ray_targets", "ray_bounds"])


def fit(model, batches, iters, log_every=0):
    """The Lightning loop the reference configures, reduced to its arithmetic: returns the list of losses."""
    (optimizer,), (sched,) = model.configure_optimizers()
    scheduler = sched["scheduler"]
    losses = []
    model.train()
    for step in range(iters):
        optimizer.zero_grad(set_to_none=True)
        out = model.training_step(next(batches), step)
        out["loss"].backward()
        nd.all_reduce_gradients(model.parameters())
        optimizer.step()
        scheduler.step()
        try:
            model.global_step = step + 1          # a LightningModule's global_step belongs to its Trainer
        except AttributeError:
            pass
        losses.append(out["loss"].detach())
        if log_every and (step + 1) % log_every == 0:
            print(f"step {step + 1}: loss {float(losses[-1]):.5f} psnr {float(mse2psnr(out['log']['train/fine_loss'])):.2f}")
    return [float(x) for x in losses]


def view_psnr(model, view, chunk=8192):
    model.eval()
    size = view["ray_directions"].shape[0]
    dirs, target = view["ray_directions"].reshape(-1, 3), view["ray_targets"].reshape(-1, 3)
    bounds = torch.tensor([model.cfg.dataset.near, model.cfg.dataset.far])
    with torch.no_grad():
        rgb = torch.cat([model.query((view["ray_origins"][None], dirs[s:s + chunk], bounds)).rgb_map
                         for s in range(0, size * size, chunk)])
    return float(mse2psnr(torch.nn.functional.mse_loss(rgb, target)))


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--iters", type=int, default=300)
    ap.add_argument("--views", type=int, default=8)
    ap.add_argument("--size", type=int, default=100, help="target images are size x size")
    ap.add_argument("--hidden-size", type=int, default=256)
    ap.add_argument("--num-layers", type=int, default=8)
    ap.add_argument("--rays", type=int, default=2048, help="cfg.nerf.train.num_random_rays")
    ap.add_argument("--lr", type=float, default=5e-4)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--save", default="", help="write a Lightning-layout checkpoint here")
    args = ap.parse_args(argv)
    from . import models
    import os
    if "RANK" in os.environ and int(os.environ.get("WORLD_SIZE", "1")) > 1:   # launched by torch.distributed.run
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        torch.distributed.init_process_group("nccl")                          # RCCL over xGMI
    rank, world = nd.world()
    device = torch.device("cuda", torch.cuda.current_device())
    flat = S.hparams(hidden_size=args.hidden_size, num_layers=args.num_layers, train_perturb=True, train_noise_std=0.0)
    flat.update({"nerf.train.num_random_rays": args.rays, "nerf.train.chunksize": args.rays, "optimizer.lr": args.lr})
    torch.manual_seed(args.seed)                                   # identical replicas on every rank
    model = models.NeRFModel(CfgNode(flat)).to(device)
    views = teacher_views(args.views + 1, args.size, device)
    held_out, train_views = views[-1], views[:-1]
    coords = torch.stack(torch.meshgrid(torch.arange(args.size), torch.arange(args.size), indexing="ij"), -1).reshape(-1, 2).to(device)
    torch.manual_seed(args.seed + 1000 * (rank + 1))               # different rays on every rank

    def batches():
        k = 0
        while True:
            yield random_ray_batch(model.cfg, train_views[k % len(train_views)], coords)
            k += 1

    before = view_psnr(model, held_out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    losses = fit(model, batches(), args.iters)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    after = view_psnr(model, held_out)
    if args.save and rank == 0:
        model.save_checkpoint(args.save)
    if rank == 0:
        print(json.dumps({"iters": args.iters, "rays_per_iter": args.rays * world, "seconds": dt,
                          "iters_per_s": args.iters / dt, "train_rays_per_s": args.iters * args.rays * world / dt,
                          "first_loss": losses[0], "last_loss": sum(losses[-10:]) / len(losses[-10:]),
                          "held_out_psnr_before": before, "held_out_psnr_after": after, "world": world}))
    return losses, before, after


if __name__ == "__main__":
    main()
