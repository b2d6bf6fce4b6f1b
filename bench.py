#!/usr/bin/env python3
"""Headline benchmark: rendered rays/sec of the NeRF ray-batch hot path on MI355X.

Workload (BASELINE.json configs[1]): nerf-synthetic-lego geometry -- 8x256 coarse + 8x256 fine
FlexibleNeRFModel, 64 coarse + 128 fine samples per ray, 800x800 views (640 000 rays), bounds [2, 6],
seeded synthetic weights / orbit poses (no dataset or checkpoint exists offline).  One "step" renders one
full view through the product path (nm_render_rays: coarse intervals -> coarse MLP -> composite ->
inverse-CDF resample -> fine MLP -> composite), ray directions already resident in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Multi-GPU: one process per GPU.  `python bench.py --gpus N` with no launcher environment re-executes itself
under `torch.distributed.run --nproc-per-node N` (and exits non-zero if the node has fewer than N GPUs); under
a launcher it reads RANK / LOCAL_RANK / WORLD_SIZE.  Views are independent -> rank r renders view
(step * N + r) (weak scaling), followed by the one real exchange step of the path, an RCCL all-gather of the
rendered pixels (7.68 MB / rank / view).

Prints ONE JSON line on rank 0 (see the task contract): metric/value, roofline of the dominant kernel
(the fused MLP; fp32 MFMA peak 157.3 TFLOP/s), the CPU baseline (the oracle = torch-CPU port of the reference
path, timed on this host's cores on a bounded ray sample), PSNR parity against it, and -- at N = 1 -- secondary
objects for BASELINE configs 4 (`mesh`: 480^3 density grid + marching cubes) and 5 (`buff`: 504x378 rays x 192
samples through the voxel-tree sampler) and one training iteration (`train`).
"""
import argparse
import json
import os
import socket
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from nerfmeshes_amd import hip_ops, synthetic as S  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
HBM_PEAK_GBS = 8000.0           # same guide, "HBM3E peak BW" (spec)
H = W = 800
NUM_COARSE, NUM_FINE = 64, 128
NEAR, FAR = 2.0, 6.0
MLP_KW = dict(num_layers=8, hidden_size=256, skip_step=4, num_encoding_fn_xyz=10, num_encoding_fn_dir=4)
PARITY_RAYS = 32768


def _pick_threads(fn, ncpu):
    """torch's default of one thread per core is far from optimal for these problem sizes on a many-core host:
    try a few thread counts on one small call each and keep the fastest."""
    best = (float("inf"), ncpu)
    for threads in sorted({min(ncpu, t) for t in (8, 16, 32, 64, ncpu)}):
        torch.set_num_threads(threads)
        fn()
        t0 = time.perf_counter()
        fn()
        dt = time.perf_counter() - t0
        if dt < best[0]:
            best = (dt, threads)
    torch.set_num_threads(best[1])
    return best[1]


def cpu_baseline(weights, rays_o, rays_d, budget_s=16.0, chunk=2048):
    """Reference path on the host cores: the oracle (a torch-CPU restatement that is bit-identical to the
    reference's NeRFModel.forward) on chunks of 2048 rays (cfg.nerf.validation.chunksize).  The rate is taken over
    the first `budget_s` seconds; the remaining rays (up to rays_d.shape[0]) are rendered untimed for the parity
    check.  Returns (rays/s, timed rays, seconds, threads, reference rgb of ALL rays)."""
    from oracle import nerf_oracle as O   # cpu_baseline leg only
    ncpu = os.cpu_count() or 1
    spec, rs = O.MLPSpec(**MLP_KW), O.RenderSpec(num_coarse=NUM_COARSE, num_fine=NUM_FINE)
    o, d = rays_o.cpu(), rays_d.cpu()
    with torch.no_grad():
        threads = _pick_threads(lambda: O.render(weights, weights, spec, spec, rs, o, d[:512], NEAR, FAR), ncpu)
        done, outs, t0 = 0, [], time.perf_counter()
        timed = None
        while done < d.shape[0]:
            _, f = O.render(weights, weights, spec, spec, rs, o, d[done:done + chunk], NEAR, FAR)
            outs.append(f["rgb_map"])
            done += min(chunk, d.shape[0] - done)
            if timed is None and time.perf_counter() - t0 >= budget_s:
                timed = (done, time.perf_counter() - t0)
        if timed is None:
            timed = (done, time.perf_counter() - t0)
    return timed[0] / timed[1], timed[0], timed[1], threads, torch.cat(outs, 0)


def _events(n):
    return [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]


def _timed(fn, reps):
    """(min ms, mean ms, last result) of `fn` with HIP events on torch's current stream (the stream every
    hip_ops wrapper launches on), after one warm-up call."""
    fn()
    torch.cuda.synchronize()
    ev, out = _events(reps), None
    for a, b in ev:
        a.record()
        out = fn()
        b.record()
    torch.cuda.synchronize()
    ms = [a.elapsed_time(b) for a, b in ev]
    return min(ms), sum(ms) / len(ms), out


def train_probe(dev, dirs, origin, rays=2048, iters=10):
    """Secondary figure: one optimizer iteration of the same 8x256 coarse+fine model on a
    2048-ray batch -- forward in train mode (perturb + noise), MSE(coarse)+MSE(fine), HIP backward, Adam."""
    from nerfmeshes_amd import models
    from nerfmeshes_amd.nerf import CfgNode
    torch.manual_seed(0)
    model = models.NeRFModel(CfgNode(S.hparams(train_perturb=True, train_noise_std=0.2))).to(dev)
    with torch.no_grad():
        for net in (model.model_coarse, model.model_fine):
            net.fc_alpha.weight.mul_(30.0)
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=5e-4)
    pick = torch.randperm(dirs.shape[0], generator=torch.Generator().manual_seed(1))[:rays].to(dev)
    batch = (origin.reshape(1, 3), dirs[pick].contiguous(), torch.tensor([2.0, 6.0]))
    target = torch.rand(rays, 3, device=dev)

    def iteration():
        opt.zero_grad(set_to_none=True)
        c, f = model(batch)
        loss = torch.nn.functional.mse_loss(c.rgb_map, target) + torch.nn.functional.mse_loss(f.rgb_map, target)
        loss.backward()
        opt.step()

    for _ in range(3):
        iteration()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        iteration()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / iters * 1e3
    return {"value": rays / ms * 1e3, "unit": "rays/s", "ms_per_iteration": ms, "rays_per_iteration": rays,
            "workload": "training step: 8x256 coarse+fine, 64+128 samples, perturb + noise, Adam (forward + HIP backward + step)"}


def mesh_probe(dev, weights, fine, res=480, limit=1.2, iso_request=32.0, cpu_points=262144):
    """BASELINE config 4 (`mesh_nerf.py --res 480 --limit 1.2 --iso-level 32`, /root/reference/src/mesh_nerf.py:27-92)
    on one GPU: the density-grid query (fused MLP, density-only trunk) against the fp32 MFMA roof, marching cubes
    against the HBM roof on its algorithmic bytes (4 B / voxel), the mesh compared bitwise IN THIS RUN with the C
    oracle on the same grid, and the CPU legs (oracle MLP on a bounded point sample; oracle marching cubes)."""
    import numpy as np
    from nerfmeshes_amd.mesh_nerf import extract_iso_level
    ax = torch.linspace(-limit, limit, res).to(dev)
    grid = torch.empty(res ** 3, dtype=torch.float32, device=dev)
    g_min, g_avg, _ = _timed(lambda: fine.grid_query(ax, ax, ax, density_only=True, out=grid), 1)
    density = grid.view(res, res, res)
    flops = res ** 3 * fine.flops_per_sample(density_only=True)

    class _A:   # the script's adaptive iso level (mesh_nerf.py:56-65)
        iso_level = iso_request
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        iso = float(extract_iso_level(density, _A))
    m_min, m_avg, (v, f, n, val) = _timed(lambda: hip_ops.marching_cubes(density, iso), 5)
    vol_bytes = res ** 3 * 4
    out = {
        "workload": f"mesh_nerf --res {res} --limit {limit} --iso-level {iso_request}: density grid + marching cubes, 1 GPU",
        "grid_query": {"points": res ** 3, "ms": g_min, "algorithmic_flops_per_point": fine.flops_per_sample(density_only=True),
                       "roofline": {"bound": "mfma", "achieved": flops / (g_min * 1e-3) / 1e12, "peak": FP32_MFMA_PEAK_TFLOPS,
                                    "unit": "TFLOP/s", "frac": flops / (g_min * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS}},
        "marching_cubes": {"iso": iso, "vertices": int(v.shape[0]), "faces": int(f.shape[0]), "ms_min": m_min, "ms_avg": m_avg,
                           "algorithmic_bytes": vol_bytes,
                           "roofline": {"bound": "hbm", "achieved": vol_bytes / (m_min * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                                        "unit": "GB/s", "frac": vol_bytes / (m_min * 1e-3) / 1e9 / HBM_PEAK_GBS},
                           "note": "whole nm_mc_count + nm_mc_emit call incl. workspace allocation and the host sync"},
    }
    # ---- CPU legs (oracle = checker + baseline)
    from oracle import mc_oracle, nerf_oracle as O
    vol = density.cpu().numpy()
    t0 = time.perf_counter()
    rv, rf, rn, rval = mc_oracle.marching_cubes(vol, iso)
    dt = time.perf_counter() - t0
    # the level itself: the GPU replays numpy's fp32 reductions (nm_np_stats) -> must equal numpy's own on the host copy
    iso_numpy = float(min(max(iso_request, vol.min() + vol.std()), vol.max() - vol.std()))
    out["marching_cubes"]["iso_equals_numpy_fp32"] = bool(iso == iso_numpy)
    same = (np.array_equal(rf, f.cpu().numpy()) and rv.tobytes() == v.cpu().numpy().tobytes()
            and rn.tobytes() == n.cpu().numpy().tobytes() and rval.tobytes() == val.cpu().numpy().tobytes())
    out["marching_cubes"].update({"bitwise_identical_to_oracle": bool(same),
                                  "cpu_baseline": {"value": dt, "unit": "s", "cores": 1, "kind": "port",
                                                   "sample": f"the full {res}^3 grid (oracle/mc_lewiner.c)"}})
    spec = O.MLPSpec(**MLP_KW)
    pts = O.grid_points(limit, res)[:: max(1, res ** 3 // cpu_points)][:cpu_points]
    with torch.no_grad():
        threads = _pick_threads(lambda: O.mlp_forward(weights, spec, pts[:8192], pts[:8192]), os.cpu_count() or 1)
        t0 = time.perf_counter()
        ref = O.mlp_forward(weights, spec, pts, pts)
        dt = time.perf_counter() - t0
    got = fine.sample_points(pts.to(dev), pts.to(dev)).cpu()
    out["grid_query"]["cpu_baseline"] = {"value": pts.shape[0] / dt, "unit": "points/s", "cores": threads,
                                         "host_cores": os.cpu_count(), "kind": "port",
                                         "sample": f"{pts.shape[0]} strided grid points, one batch, {dt:.2f} s"}
    out["grid_query"]["speedup_vs_cpu"] = (res ** 3 / (g_min * 1e-3)) / (pts.shape[0] / dt)
    scale = float(ref[:, 3].abs().max()) + 1.0
    out["grid_query"]["parity"] = {"max_abs_dsigma_over_scale": float((got[:, 3] - ref[:, 3]).abs().max()) / scale,
                                   "max_abs_drgb": float((got[:, :3] - ref[:, :3]).abs().max()), "points": int(pts.shape[0])}
    return out


def buff_probe(dev, cpu_rays=2048):
    """BASELINE config 5 geometry (/root/reference/config/buff-colmap-fern.yml:31-74): BuFFModel.query on a
    504x378 view (fern 4032x3024 / 8), 192 samples per ray placed by the voxel-tree sampler (12^3 voxels on
    [-0.6, 0.6]^3), single 8x256 network, bounds [0, 1.2], synthetic pose on radius 1."""
    from nerfmeshes_amd import models
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

    def view():
        outs = []
        for s in range(0, d.shape[0], chunk):
            outs.append(model.query((o[None], d[s:s + chunk], bounds)).rgb_map)
        return torch.cat(outs, 0)

    with torch.no_grad():
        hip_ops.mlp_profile_enable(True)
        view()
        torch.cuda.synchronize()
        hip_ops.mlp_profile_read()
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            rgb = view()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / reps
        launches, kernel_ms, kernel_flops = hip_ops.mlp_profile_read()
        hip_ops.mlp_profile_enable(False)
        i_min, i_avg, (z, idx, mask) = _timed(
            lambda: model.tree.batch_ray_voxel_intersect(o[None], d[:chunk], 0.0, 1.2, 192), 5)
    rays = hh * ww
    achieved = kernel_flops / (kernel_ms * 1e-3) / 1e12 if kernel_ms > 0 else 0.0
    out = {
        "workload": "buff-colmap-fern geometry: BuFFModel.query, 504x378 rays x 192 tree-placed samples, 8x256 network, 1 GPU",
        "value": rays / wall, "unit": "rays/s", "ms_per_view": wall * 1e3, "rays_per_view": rays, "chunk_rays": chunk,
        "algorithmic_flops_per_ray": 192 * model.model.hip().flops_per_sample(),
        "roofline": {"bound": "mfma", "achieved": achieved, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": achieved / FP32_MFMA_PEAK_TFLOPS, "launches": launches,
                     "mlp_kernel_share_of_wall": kernel_ms * 1e-3 / (wall * reps)},
        "nm_buff_intersect": {"ms_min": i_min, "ms_avg": i_avg, "rays": chunk, "voxels": int(model.tree.voxels.shape[0]),
                              "samples": 192, "rays_hitting_tree": float(mask.float().mean())},
    }
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


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--chunk", type=int, default=65536, help="rays per nm_render_rays call")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train-probe", action="store_true")
    ap.add_argument("--no-mesh-probe", action="store_true")
    ap.add_argument("--no-buff-probe", action="store_true")
    ap.add_argument("--no-b3-probe", action="store_true")
    ap.add_argument("--headline-only", action="store_true", help="skip every secondary object and the CPU legs")
    args = ap.parse_args()
    if args.headline_only:
        args.no_cpu_baseline = args.no_train_probe = args.no_mesh_probe = args.no_buff_probe = args.no_b3_probe = True

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a MI355X (no CPU fallback exists for the hot path)")
    visible = torch.cuda.device_count()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > visible:
        raise SystemExit(f"--gpus {args.gpus} but only {visible} GPU(s) are visible on this node")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher: become one.  One process per GPU over RCCL, rendezvous on 127.0.0.1.
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        os.execv(sys.executable, cmd)

    # stdout carries the ONE JSON line and nothing else: whatever native libraries print there (RCCL's version banner
    # at communicator creation / teardown, which lands AFTER the line) is sent to stderr by pointing fd 1 at fd 2 and
    # keeping the real stdout for the final write
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    # a launcher environment (WORLD_SIZE set, also WORLD_SIZE=1) or NM_BENCH_FORCE_DIST=1 exercises the RCCL path
    use_dist = world > 1 or "WORLD_SIZE" in os.environ or os.environ.get("NM_BENCH_FORCE_DIST") == "1"
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(_free_port()) if world == 1 else "29500")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    weights = S.make_scene_weights(**MLP_KW)
    coarse = hip_ops.HipMLP(weights, MLP_KW, dev)
    fine = hip_ops.HipMLP(weights, MLP_KW, dev)
    u_c = torch.linspace(0.0, 1.0, NUM_COARSE).to(dev)
    u_f = torch.linspace(0.0, 1.0, NUM_FINE).to(dev)
    near, far = torch.tensor([NEAR], device=dev), torch.tensor([FAR], device=dev)

    total_steps = args.warmup + args.steps
    poses = S.orbit_poses(max(total_steps * world, 1))
    views = []   # ray directions resident in HBM before the timed region
    for s in range(total_steps):
        o, d = hip_ops.ray_bundle(poses[s * world + rank], H, W, S.LEGO_FOCAL_800, device=dev)
        views.append((o[None].contiguous(), d))
    image = torch.empty(H * W, 3, device=dev)
    gathered = torch.empty(world * H * W, 3, device=dev) if use_dist else None

    def step(i):
        o, d = views[i]
        for s in range(0, H * W, args.chunk):
            _, fb = hip_ops.render_rays(coarse, fine, o, d[s:s + args.chunk], near, far, u_c, u_f)
            image[s:s + args.chunk] = fb["rgb_map"]
        if use_dist:
            dist.all_gather_into_tensor(gathered, image)

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    fence()
    hip_ops.mlp_profile_enable(True)
    hip_ops.mlp_profile_read()
    t0 = time.perf_counter()
    for i in range(args.warmup, total_steps):
        step(i)
    fence()
    elapsed = time.perf_counter() - t0
    launches, kernel_ms, kernel_flops = hip_ops.mlp_profile_read()
    hip_ops.mlp_profile_enable(False)
    achieved = kernel_flops / (kernel_ms * 1e-3) / 1e12 if kernel_ms > 0 else 0.0
    rccl = None
    if use_dist:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # what the all-gather actually delivered: every rank's slot of the last step must hold THAT rank's pixels
        # (slot checksums are compared with the checksums the ranks computed locally), and per-rank roofline fractions
        mine = torch.stack([image.double().sum(), torch.tensor(achieved, device=dev, dtype=torch.float64)])
        allv = torch.empty(world, 2, device=dev, dtype=torch.float64)
        dist.all_gather_into_tensor(allv, mine)
        slots = gathered.view(world, H * W, 3).double().sum(dim=(1, 2))
        rccl = {"backend": dist.get_backend(), "ranks_in_all_gather": int(dist.get_world_size()),
                "gathered_bytes_per_step": int(gathered.numel() * 4),
                "slots_match_rank_checksums": bool(torch.allclose(slots, allv[:, 0], rtol=1e-9, atol=0)),   # fp64 sums, two reduction shapes
                "roofline_frac_per_rank": [float(x) / FP32_MFMA_PEAK_TFLOPS for x in allv[:, 1]]}

    rays_total = args.steps * H * W * world
    value = rays_total / elapsed
    flops_per_ray = (NUM_COARSE + NUM_COARSE + NUM_FINE) * coarse.flops_per_sample()

    traffic = None
    for name in ("r02_pmc_mlp_kernel.json", "r01_pmc_mlp_kernel.json"):
        pmc = os.path.join(ROOT, "profiles", name)
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
                break
            except Exception:
                traffic = None

    out = {
        "metric": "rendered rays/sec (64+128 samples, 8x256 MLP), lego scene geometry",
        "value": value, "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "nerf-synthetic lego: 8x256 coarse+fine MLP, 64 coarse + 128 fine samples, "
                               "800x800 view per step per GPU, bounds [2,6], seeded weights/orbit poses",
                   "rays_per_step_per_gpu": H * W, "chunk_rays": args.chunk,
                   "parallelism": f"views sharded over {world} GPU(s), RCCL all-gather of pixels" if world > 1
                   else "single GPU"},
        "roofline": {"bound": "mfma", "achieved": achieved, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": achieved / FP32_MFMA_PEAK_TFLOPS, "traffic": traffic,
                     "kernel": "nm::mlp_kernel3<256,10,4,8,8,1>", "launches": launches,
                     "avg_launch_ms": kernel_ms / max(launches, 1),
                     "algorithmic_flops_per_ray": flops_per_ray,
                     "mlp_kernel_share_of_wall": kernel_ms * 1e-3 / elapsed},
    }
    if rccl is not None:
        out["rccl"] = rccl

    solo = rank == 0 and world == 1
    ref_idx = ref_rgb = None
    if solo and not args.no_cpu_baseline:
        from oracle import parity
        o, d = views[0]
        idx = torch.arange(0, H * W, (H * W) // PARITY_RAYS, device=dev)[:PARITY_RAYS]   # strided sample of the view
        rps, n_timed, dt, threads, ref_rgb = cpu_baseline(weights, o, d[idx])
        ref_idx = idx
        _, fb = hip_ops.render_rays(coarse, fine, o, d[idx].contiguous(), near, far, u_c, u_f)
        out["cpu_baseline"] = {"value": rps, "unit": "rays/s", "cores": threads, "host_cores": os.cpu_count(),
                               "kind": "port",
                               "sample": f"{n_timed} rays of view 0 (stride {(H * W) // PARITY_RAYS}), chunks of 2048, {dt:.1f} s "
                                         f"({threads} torch threads = fastest of 8/16/32/64/{os.cpu_count()} on this host)",
                               "speedup": value / rps}
        out["parity"] = parity.psnr_parity(fb["rgb_map"].cpu(), ref_rgb, chunk=2048)
    for name, skip, fn in (("train", args.no_train_probe, lambda: train_probe(dev, views[0][1], views[0][0])),
                           ("mesh", args.no_mesh_probe, lambda: mesh_probe(dev, weights, fine)),
                           ("buff", args.no_buff_probe, lambda: buff_probe(dev)),
                           ("bf16x3", args.no_b3_probe,
                            lambda: b3_probe(dev, weights, views, near, far, u_c, u_f, args.chunk, ref_idx, ref_rgb))):
        if solo and not skip:
            try:
                out[name] = fn()
            except Exception as e:  # the headline line must not depend on a secondary figure
                out[name] = {"error": repr(e)}
    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
