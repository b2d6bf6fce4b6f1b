"""bench.py's other secondary objects: `buff` (config 5), `eval` (config 3 at N = 1), `tiny` (config 1), `bf16x3` (the opt-in precision)."""
import os
import time

import torch

from nerfmeshes_amd import hip_ops, synthetic as S

from .common import (FAR, FP32_MFMA_PEAK_TFLOPS, H, MLP_KW, NEAR, NUM_COARSE, NUM_FINE, W, _per_rank, _pick_threads, _timed,
                     _wall_max, cpu_baseline)
from .train import tiny_train_probe


def buff_probe(dev, cpu_rays=2048, rank=0, world=1, use_dist=False, cpu_legs=True):
    """BASELINE config 5 geometry (/root/reference/config/buff-colmap-fern.yml:31-74): BuFFModel.query on a
    504x378 view (fern 4032x3024 / 8), 192 samples per ray placed by the voxel-tree sampler (12^3 voxels on
    [-0.6, 0.6]^3), single 8x256 network, bounds [0, 1.2], synthetic pose on radius 1.  At N > 1 the view's rays are
    split into contiguous ranges over the ranks and the pixels all-gathered (strong scaling of one view)."""
    from nerfmeshes_amd import dist as nd, models
    hp = S.hparams(model="BuFFModel", use_fine=False, num_coarse=192, num_fine=64, near=0.0, far=1.2, dataset_type="colmap")
    w = S.make_mlp_weights(9, density_gain=1500.0, density_bias=60.0, **MLP_KW)
    model = models.BuFFModel(hp)
    sd = model.state_dict()
    for k, v in w.items():
        sd["model." + k] = torch.from_numpy(v)
    model.load_state_dict(sd)
    model = model.eval().to(dev)
    hh, ww = 378, 504
    o, d = hip_ops.ray_bundle(S.pose_spherical(30.0, -20.0, 1.0), hh, ww, 0.8 * ww, device=dev)
    bounds = torch.tensor([0.0, 1.2])
    chunk = 65536
    rays = hh * ww
    lo, hi = nd.split_range(rays, rank, world)
    counts = [b_ - a_ for a_, b_ in (nd.split_range(rays, r, world) for r in range(world))]

    def view():
        outs = []
        for s0 in range(lo, hi, chunk):
            outs.append(model.query((o[None], d[s0:min(s0 + chunk, hi)], bounds)).rgb_map)
        mine = torch.cat(outs, 0)
        return nd.all_gather_rows(mine, counts) if world > 1 else mine

    with torch.no_grad():
        hip_ops.mlp_profile_enable(True)
        view()
        torch.cuda.synchronize()
        hip_ops.mlp_profile_read()
        reps = 3
        wall, rgb = _wall_max(lambda: [view() for _ in range(reps)][-1], dev, use_dist)
        wall /= reps
        launches, kernel_ms, kernel_flops = hip_ops.mlp_profile_read()
        hip_ops.mlp_profile_enable(False)
        timings = {}
        for tie in ("stable", "reference"):
            model.tree.tie_order = tie
            t_min, t_avg, (z, idx, mask) = _timed(lambda: model.tree.batch_ray_voxel_intersect(o[None], d[:chunk], 0.0, 1.2, 192), 5)
            timings[tie] = {"ms_min": t_min, "ms_avg": t_avg}
        model.tree.tie_order = "auto"
    achieved = kernel_flops / (kernel_ms * 1e-3) / 1e12 if kernel_ms > 0 else 0.0
    out = {
        "workload": "buff-colmap-fern geometry: BuFFModel.query, 504x378 rays x 192 tree-placed samples, 8x256 network, "
                    + (f"rays split over {world} ranks + all-gather of the pixels" if world > 1 else "1 GPU"),
        "value": rays / wall, "unit": "rays/s", "ms_per_view": wall * 1e3, "rays_per_view": rays, "rays_per_rank": counts, "chunk_rays": chunk,
        "scaling": "strong" if world > 1 else None,
        "algorithmic_flops_per_ray": 192 * model.model.hip().flops_per_sample(),
        "roofline": {"bound": "mfma", "achieved": achieved, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": achieved / FP32_MFMA_PEAK_TFLOPS, "launches": launches,
                     "frac_per_rank": _per_rank(achieved / FP32_MFMA_PEAK_TFLOPS, dev, world, use_dist),
                     "mlp_kernel_share_of_wall": kernel_ms * 1e-3 / (wall * reps)},
        "nm_buff_intersect": {"rays": chunk, "voxels": int(model.tree.voxels.shape[0]), "samples": 192,
                              "rays_hitting_tree": float(mask.float().mean()),
                              "tie_order_stable": timings["stable"], "tie_order_reference": timings["reference"],
                              "note": "stable: every id is the voxel its sample lies in (eval default); reference: the reference's "
                                      "own ids (its three unstable sorts replayed), the default while training"},
    }
    if rank != 0 or not cpu_legs:
        return out
    # ---- CPU leg: the oracle's BuFF chain on a bounded strided ray sample + parity on those rays
    from oracle import nerf_oracle as O, parity
    spec, rs = O.MLPSpec(**MLP_KW), O.RenderSpec(num_coarse=192, num_fine=0)
    pick = torch.arange(0, rays, max(1, rays // cpu_rays), device=dev)[:cpu_rays]
    dd = d[pick].contiguous()
    vox = model.tree.voxels.detach().cpu()

    def cpu_render(dirs):
        zz, _, mm = O.buff_intersect(vox, o[None].cpu(), dirs, 0.0, 1.2, 192)
        uni = O.coarse_intervals(0.0, 1.2, 192, dirs.shape[0]).contiguous()
        zz = torch.where(mm[:, None], zz, uni)
        pts = O.ray_points(zz, dirs, o[None].cpu()).reshape(-1, 3)
        rad = O.mlp_forward(w, spec, pts, dirs[:, None, :].expand(-1, 192, -1).reshape(-1, 3)).reshape(dirs.shape[0], 192, 4)
        return O.composite(rad, zz, dirs, rs)["rgb_map"]

    with torch.no_grad():
        dc = dd.cpu()
        threads = _pick_threads(lambda: cpu_render(dc[:256]), os.cpu_count() or 1)
        t0 = time.perf_counter()
        ref = torch.cat([cpu_render(dc[s:s + 1024]) for s in range(0, dc.shape[0], 1024)], 0)
        dt = time.perf_counter() - t0
        got = model.query((o[None], dd, bounds)).rgb_map.cpu()
    out["cpu_baseline"] = {"value": dc.shape[0] / dt, "unit": "rays/s", "cores": threads, "host_cores": os.cpu_count(),
                           "kind": "port", "sample": f"{dc.shape[0]} strided rays of the view, chunks of 1024, {dt:.1f} s"}
    out["speedup_vs_cpu"] = out["value"] / out["cpu_baseline"]["value"]
    out["parity"] = parity.psnr_parity(got, ref, chunk=1024)
    return out


def eval_probe(dev, weights, views=20, render_chunk=65536, cpu_views=4, cpu_size=92, cpu_legs=True):
    """BASELINE config 3 at N = 1 (`eval_nerf.py` over a test set, /root/reference/src/eval_nerf.py:50-105): `views` orbit
    views of 800x800 through the eval_nerf mirror (`eval_views`: per-view loss = sum of per-2048-ray-chunk MSEs divided by
    the FLOAT batch count 312.5, dataset loss = mean over views, PSNR of that), every view scored against a seeded noisy
    photograph of itself (~34 dB, the regime a trained NeRF is scored in).  Parity leg: `cpu_views` small views rendered by
    the oracle on the host, scored by the oracle's bookkeeping, against the same views through the mirror."""
    import contextlib, io
    from nerfmeshes_amd import eval_nerf as E, models
    from nerfmeshes_amd.nerf import CfgNode
    from nerfmeshes_amd.models.model_helpers import nest_dict
    hp = S.hparams()
    model = models.NeRFModel(hp)
    sd = model.state_dict()
    for k, v in weights.items():
        sd["model_coarse." + k] = torch.from_numpy(v)
        sd["model_fine." + k] = torch.from_numpy(v)
    model.load_state_dict(sd)
    model = model.eval().to(dev)
    cfg = CfgNode(nest_dict(hp, sep="."))
    gen = torch.Generator(device=dev)

    def photograph(view_nr, rgb):       # the view's own render + seeded noise, clamped: PSNR ~ 34 dB by construction
        gen.manual_seed(1000 + view_nr)
        return (rgb + 0.02 * torch.randn(rgb.shape, generator=gen, device=dev)).clamp_(0.0, 1.0)

    def run(n):
        vs = [(pose, H, W, S.LEGO_FOCAL_800, photograph) for pose in S.orbit_poses(n)]
        with contextlib.redirect_stdout(io.StringIO()), torch.no_grad():
            return E.eval_views(model, vs, cfg, dev, render_chunk=render_chunk)

    run(1)
    torch.cuda.synchronize()
    hip_ops.mlp_profile_enable(True)
    hip_ops.mlp_profile_read()
    t0 = time.perf_counter()
    losses, total, psnr, _ = run(views)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    launches, kernel_ms, kernel_flops = hip_ops.mlp_profile_read()
    hip_ops.mlp_profile_enable(False)
    achieved = kernel_flops / (kernel_ms * 1e-3) / 1e12 if kernel_ms > 0 else 0.0
    # what the UNMODIFIED script's loop shape costs: every call is one cfg.nerf.validation.chunksize = 2048-ray chunk
    # (eval_nerf.py:62-65), i.e. 313 calls of ~8 launches per view instead of 10
    vs1 = [(S.orbit_poses(views)[0], H, W, S.LEGO_FOCAL_800, photograph)]
    with contextlib.redirect_stdout(io.StringIO()), torch.no_grad():
        E.eval_views(model, vs1, cfg, dev)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        small_losses, _, _, _ = E.eval_views(model, vs1, cfg, dev)
        torch.cuda.synchronize()
        wall_2048 = time.perf_counter() - t1
    out = {"workload": f"config 3 at N = 1: {views} orbit views of 800x800 through the eval_nerf mirror (eval_views), 8x256 coarse+fine, 64+128, "
                       f"rays generated in the kernels, rendered in calls of {render_chunk} rays, loss bookkeeping per 2048 rays / float batch_count 312.5",
           "value": views * H * W / wall, "unit": "rays/s", "views": views, "ms_per_view": wall / views * 1e3,
           "dataset_loss_mse": float(total), "dataset_psnr_db": float(psnr),
           "per_view_psnr_db_min_max": [float(min(-10.0 * torch.log10(l) for l in losses)), float(max(-10.0 * torch.log10(l) for l in losses))],
           "roofline": {"bound": "mfma", "achieved": achieved, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": achieved / FP32_MFMA_PEAK_TFLOPS, "launches": launches,
                        "mlp_kernel_share_of_wall": kernel_ms * 1e-3 / wall},
           "at_reference_chunksize": {"chunk_rays": int(cfg.nerf.validation.chunksize), "value": H * W / wall_2048, "unit": "rays/s",
                                      "ms_per_view": wall_2048 * 1e3, "same_loss_as_large_calls": bool(float(small_losses[0]) == float(losses[0])),
                                      "note": "one view rendered in the reference's own 2048-ray calls (313 per view): what the unmodified "
                                              "eval_nerf.py loop gets without raising nerf.validation.chunksize"}}
    if not cpu_legs:
        return out
    from oracle import nerf_oracle as O, parity
    spec, rs = O.MLPSpec(**MLP_KW), O.RenderSpec(num_coarse=NUM_COARSE, num_fine=NUM_FINE)
    focal = S.LEGO_FOCAL_800 * cpu_size / 800.0
    small, ref_losses = [], []
    t0 = time.perf_counter()
    with torch.no_grad():
        for i, pose in enumerate(S.orbit_poses(views)[:: max(1, views // cpu_views)][:cpu_views]):
            o, d = O.get_ray_bundle(cpu_size, cpu_size, focal, torch.from_numpy(pose))
            d = d.reshape(-1, 3)
            ref = torch.cat([O.render(weights, weights, spec, spec, rs, o[None], d[s0:s0 + 2048], NEAR, FAR)[1]["rgb_map"]
                             for s0 in range(0, d.shape[0], 2048)])
            tgt = parity.noisy_targets(ref, seed=parity.TARGET_SEED + i)
            ref_losses.append(O.view_loss(ref, tgt, 2048))
            small.append((pose, cpu_size, cpu_size, focal, tgt))
    dt = time.perf_counter() - t0
    ref_total = O.dataset_loss(ref_losses)
    with contextlib.redirect_stdout(io.StringIO()), torch.no_grad():
        got_losses, got_total, got_psnr, _ = E.eval_views(model, small, cfg, dev)
    rays = cpu_views * cpu_size * cpu_size
    out["cpu_baseline"] = {"value": rays / dt, "unit": "rays/s", "cores": torch.get_num_threads(), "host_cores": os.cpu_count(), "kind": "port",
                           "sample": f"{cpu_views} views of {cpu_size}x{cpu_size} ({rays} rays; batch_count {cpu_size * cpu_size / 2048}), chunks of 2048, {dt:.1f} s"}
    out["parity"] = {"views": cpu_views, "rays_per_view": cpu_size * cpu_size, "float_batch_count": cpu_size * cpu_size / 2048,
                     "dataset_psnr_ref_db": float(O.mse2psnr(ref_total)), "dataset_psnr_hip_db": float(got_psnr),
                     "abs_dpsnr_db": abs(float(O.mse2psnr(ref_total)) - float(got_psnr)),
                     "per_view_abs_dpsnr_db": [abs(float(O.mse2psnr(a)) - float(O.mse2psnr(b.cpu()))) for a, b in zip(ref_losses, got_losses)],
                     "targets": "oracle render + N(0,0.02) PCG64, per view; HIP and oracle scored against the same targets by their own bookkeeping"}
    return out


def b3_probe(dev, weights, views, near, far, u_c, u_f, chunk, ref_idx, ref_rgb):
    """Opt-in precision mode "bf16x3" (every fp32 product emulated by six bf16 MFMA products of three-way operand
    splits, fp32 accumulation) on the headline workload: one 800x800 view, and its own PSNR parity against the SAME CPU
    reference render the fp32 path is scored on.  fp32 stays the default and the headline dtype."""
    from oracle import parity
    b3 = hip_ops.HipMLP(weights, MLP_KW, dev, precision="bf16x3")
    o, d = views[0]

    def view():
        for s in range(0, H * W, chunk):
            hip_ops.render_rays(b3, b3, o, d[s:s + chunk], near, far, u_c, u_f)

    hip_ops.mlp_profile_enable(True)
    view()
    torch.cuda.synchronize()
    hip_ops.mlp_profile_read()
    t0 = time.perf_counter()
    view()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    launches, kernel_ms, kernel_flops = hip_ops.mlp_profile_read()
    hip_ops.mlp_profile_enable(False)
    out = {"workload": "the headline view through the opt-in bf16x3 kernels (fp32-emulating: 3-way bf16 split of both "
                       "operands, 6 bf16 MFMA products, fp32 accumulation)",
           "value": H * W / wall, "unit": "rays/s", "ms_per_view": wall * 1e3, "dtype": "bf16x3",
           "algorithmic_tflops": kernel_flops / (kernel_ms * 1e-3) / 1e12 if kernel_ms > 0 else 0.0,
           "note": "algorithmic FLOP of the fp32 network / kernel time; the bf16 matrix pipe executes 6x as many"}
    if ref_rgb is not None:
        _, fb = hip_ops.render_rays(b3, b3, o, d[ref_idx].contiguous(), near, far, u_c, u_f)
        out["parity"] = parity.psnr_parity(fb["rgb_map"].cpu(), ref_rgb, chunk=2048)
    return out


def tiny_probe(dev, cpu_legs=True):
    """BASELINE config 1 (`config/tiny.yaml` sizes: 4-layer x 64 MLP, 32 coarse samples, no fine network, ONE 400x400 view;
    the reference runs it on the CPU as plumbing): the same product path on the GPU, its MLP kernel against the fp32 MFMA
    roof (useful FLOP only -- the 64-wide layers pad their encodings), parity against the oracle on a ray sample, and the
    oracle timed on the host."""
    kw = dict(num_layers=4, hidden_size=64, skip_step=4, num_encoding_fn_xyz=6, num_encoding_fn_dir=4)
    w = S.make_mlp_weights(11, density_gain=30.0, density_bias=0.3, **kw)
    net = hip_ops.HipMLP(w, kw, dev)
    hh = ww = 400
    o, d = hip_ops.ray_bundle(S.orbit_poses(4)[1], hh, ww, S.LEGO_FOCAL_800 / 2, device=dev)
    near, far = torch.tensor([NEAR], device=dev), torch.tensor([FAR], device=dev)
    u_c = torch.linspace(0.0, 1.0, 32).to(dev)

    def view():
        return hip_ops.render_rays(net, None, o[None], d, near, far, u_c, None)[0]["rgb_map"]

    hip_ops.mlp_profile_enable(True)
    view()
    torch.cuda.synchronize()
    hip_ops.mlp_profile_read()
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        rgb = view()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / reps
    launches, kernel_ms, kernel_flops = hip_ops.mlp_profile_read()
    hip_ops.mlp_profile_enable(False)
    achieved = kernel_flops / (kernel_ms * 1e-3) / 1e12 if kernel_ms > 0 else 0.0
    out = {"workload": "config 1 (tiny): 4x64 MLP, 32 coarse samples, no fine network, one 400x400 view, 1 GPU",
           "value": hh * ww / wall, "unit": "rays/s", "ms_per_view": wall * 1e3, "rays_per_view": hh * ww,
           "algorithmic_flops_per_ray": 32 * net.flops_per_sample(),
           "roofline": {"bound": "mfma", "achieved": achieved, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": achieved / FP32_MFMA_PEAK_TFLOPS, "launches": launches,
                        "mlp_kernel_share_of_wall": kernel_ms * 1e-3 / (wall * reps),
                        "note": "one 160 000-ray launch of 5.1 M samples lasts ~2 ms: the share of wall not in the MLP kernel is the per-ray kernels and launch latency"}}
    if not cpu_legs:
        return out
    from oracle import nerf_oracle as O, parity
    spec, rs = O.MLPSpec(**kw), O.RenderSpec(num_coarse=32, num_fine=0)
    idx = torch.arange(0, hh * ww, 10, device=dev)                       # 16 000 rays
    dc = d[idx].cpu()
    with torch.no_grad():
        threads = _pick_threads(lambda: O.render(w, None, spec, None, rs, o[None].cpu(), dc[:2048], NEAR, FAR), os.cpu_count() or 1)
        t0 = time.perf_counter()
        ref = torch.cat([O.render(w, None, spec, None, rs, o[None].cpu(), dc[s0:s0 + 2048], NEAR, FAR)[0]["rgb_map"]
                         for s0 in range(0, dc.shape[0], 2048)])
        dt = time.perf_counter() - t0
    out["cpu_baseline"] = {"value": dc.shape[0] / dt, "unit": "rays/s", "cores": threads, "host_cores": os.cpu_count(), "kind": "port",
                           "sample": f"{dc.shape[0]} rays of the view (stride 10), chunks of 2048, {dt:.2f} s"}
    out["speedup_vs_cpu"] = out["value"] / out["cpu_baseline"]["value"]
    out["parity"] = parity.psnr_parity(rgb[idx].cpu(), ref, chunk=2048)
    try:
        out["train"] = tiny_train_probe(dev)
    except Exception as e:      # a figure of a figure: never at the expense of the rest of the object
        out["train"] = {"error": f"{type(e).__name__}: {e}"}
    return out
