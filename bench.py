#!/usr/bin/env python3
"""Headline benchmark: rendered rays/sec of the NeRF ray-batch hot path on MI355X.

Workload (BASELINE.json configs[1]): nerf-synthetic-lego geometry -- 8x256 coarse + 8x256 fine
FlexibleNeRFModel, 64 coarse + 128 fine samples per ray, 800x800 views (640 000 rays), bounds [2, 6],
seeded synthetic weights / orbit poses (no dataset or checkpoint exists offline).  One "step" renders one
full view through the product path (nm_render_rays: coarse intervals -> coarse MLP -> composite ->
inverse-CDF resample -> fine MLP -> composite), ray directions already resident in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Multi-GPU: views are independent -> rank r renders view (step * N + r) (weak scaling), followed by the
one real exchange step of the path, an RCCL all-gather of the rendered pixels (7.68 MB / rank / view).

Prints ONE JSON line on rank 0 (see the task contract): metric/value, roofline of the dominant kernel
(the fused MLP; fp32 MFMA peak 157.3 TFLOP/s), and the CPU baseline (the oracle = torch-CPU port of
the reference path, timed on this host's cores on a bounded ray sample).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from nerfmeshes_amd import hip_ops, synthetic as S  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
H = W = 800
NUM_COARSE, NUM_FINE = 64, 128
NEAR, FAR = 2.0, 6.0
MLP_KW = dict(num_layers=8, hidden_size=256, skip_step=4, num_encoding_fn_xyz=10, num_encoding_fn_dir=4)


def cpu_baseline(weights, rays_o, rays_d, budget_s=16.0, chunk=2048):
    """Reference path on the host cores: the oracle (a torch-CPU restatement that is bit-identical to the
    reference's NeRFModel.forward) on chunks of 2048 rays (cfg.nerf.validation.chunksize).  torch's
    default of one thread per core is far from optimal for this problem size on a many-core host, so a
    few thread counts are tried on one chunk each and the fastest is used for the timed sample."""
    from oracle import nerf_oracle as O   # cpu_baseline leg only
    ncpu = os.cpu_count() or 1
    spec, rs = O.MLPSpec(**MLP_KW), O.RenderSpec(num_coarse=NUM_COARSE, num_fine=NUM_FINE)
    o, d = rays_o.cpu(), rays_d.cpu()
    best = (0.0, ncpu)
    with torch.no_grad():
        for threads in sorted({min(ncpu, t) for t in (8, 16, 32, 64, ncpu)}):
            torch.set_num_threads(threads)
            O.render(weights, weights, spec, spec, rs, o, d[:256], NEAR, FAR)      # warm-up
            t0 = time.perf_counter()
            O.render(weights, weights, spec, spec, rs, o, d[:512], NEAR, FAR)
            rate = 512 / (time.perf_counter() - t0)
            if rate > best[0]:
                best = (rate, threads)
        torch.set_num_threads(best[1])
        done, outs, t0 = 0, [], time.perf_counter()
        while done < d.shape[0] and (time.perf_counter() - t0 < budget_s or done < chunk):
            _, f = O.render(weights, weights, spec, spec, rs, o, d[done:done + chunk], NEAR, FAR)
            outs.append(f["rgb_map"])
            done += min(chunk, d.shape[0] - done)
        dt = time.perf_counter() - t0
    return done / dt, done, dt, torch.cat(outs, 0), O


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--chunk", type=int, default=65536, help="rays per nm_render_rays call")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train-probe", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a MI355X (no CPU fallback exists for the hot path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    force_dist = os.environ.get("NM_BENCH_FORCE_DIST") == "1"   # exercise the RCCL path on a 1-GPU box
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
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
    use_dist = dist is not None
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
    if use_dist:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    rays_total = args.steps * H * W * world
    value = rays_total / elapsed
    achieved = kernel_flops / (kernel_ms * 1e-3) / 1e12 if kernel_ms > 0 else 0.0
    flops_per_ray = (NUM_COARSE + NUM_COARSE + NUM_FINE) * coarse.flops_per_sample()

    traffic = None
    pmc = os.path.join(ROOT, "profiles", "r01_pmc_mlp_kernel.json")
    if os.path.exists(pmc):
        try:
            traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
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
                     "kernel": "nm::mlp_kernel<256,10,4,8>", "launches": launches,
                     "avg_launch_ms": kernel_ms / max(launches, 1),
                     "algorithmic_flops_per_ray": flops_per_ray,
                     "mlp_kernel_share_of_wall": kernel_ms * 1e-3 / elapsed},
    }

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        o, d = views[0]
        idx = torch.arange(0, H * W, (H * W) // 32768, device=dev)[:32768]       # strided sample of the view
        rps, n, dt, ref_rgb, O = cpu_baseline(weights, o, d[idx])
        _, fb = hip_ops.render_rays(coarse, fine, o, d[idx[:n]].contiguous(), near, far, u_c, u_f)
        got = fb["rgb_map"].cpu()
        tgt = torch.from_numpy(S.pseudo_targets(n))
        p_ref = float(O.mse2psnr(O.view_loss(ref_rgb, tgt, 2048)))
        p_got = float(O.mse2psnr(O.view_loss(got, tgt, 2048)))
        out["cpu_baseline"] = {"value": rps, "unit": "rays/s", "cores": torch.get_num_threads(), "kind": "port",
                               "sample": f"{n} rays of view 0 (stride {(H * W) // 32768}), chunks of 2048, {dt:.1f} s",
                               "speedup": value / rps}
        out["parity"] = {"psnr_ref_db": p_ref, "psnr_hip_db": p_got, "abs_dpsnr_db": abs(p_ref - p_got),
                         "max_abs_drgb": float((got - ref_rgb).abs().max()),
                         "psnr_hip_vs_ref_db": float(O.mse2psnr(torch.nn.functional.mse_loss(got, ref_rgb))),
                         "rays": n}
    if rank == 0 and world == 1 and not args.no_train_probe:
        try:
            o, d = views[0]
            out["train"] = train_probe(dev, d, o)
        except Exception as e:  # the headline line must not depend on the secondary figure
            out["train"] = {"error": repr(e)}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
