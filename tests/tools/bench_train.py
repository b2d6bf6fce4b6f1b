"""Training-step benchmark: one optimizer iteration of NeRFModel on a 2048-ray batch,
lego shape (8x256, 64 coarse + 128 fine), through the public API: forward (train mode: perturb + noise) ->
MSE(coarse) + MSE(fine) -> backward -> Adam step.  Prints one JSON object with the iteration time, a per-stage
breakdown (HIP events on torch's stream) and -- bounded -- the same iteration through torch autograd over the
CPU oracle on the host cores.

    python tests/tools/bench_train.py [--rays 2048] [--iters 20] [--cpu-rays 256]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

from nerfmeshes_amd import models, synthetic as S, train_ops as T  # noqa: E402
from nerfmeshes_amd.nerf import CfgNode  # noqa: E402


def timed(fn, iters):
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=2048)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--cpu-rays", type=int, default=256)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    cfg = CfgNode(S.hparams(train_perturb=True, train_noise_std=0.2))
    torch.manual_seed(0)
    model = models.NeRFModel(cfg).to(dev)
    with torch.no_grad():
        for net in (model.model_coarse, model.model_fine):
            net.fc_alpha.weight.mul_(30.0)
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=5e-4)
    R = args.rays
    pose = S.pose_spherical(30.0, -30.0, 4.0)
    from nerfmeshes_amd import hip_ops
    origin, dirs = hip_ops.ray_bundle(pose, 800, 800, 1111.1111, 0, 640000, dev)
    pick = torch.randperm(640000, generator=torch.Generator().manual_seed(1))[:R].to(dev)
    batch = (origin[None], dirs[pick].contiguous(), torch.tensor([2.0, 6.0]))
    target = torch.rand(R, 3, device=dev)

    def iteration():
        opt.zero_grad(set_to_none=True)
        c, f = model(batch)
        loss = torch.nn.functional.mse_loss(c.rgb_map, target) + torch.nn.functional.mse_loss(f.rgb_map, target)
        loss.backward()
        opt.step()
        return loss

    for _ in range(3):
        iteration()
    ms_iter = timed(iteration, args.iters)

    # ---- stage breakdown on the fine network's shapes (192 samples / ray)
    fine = model.model_fine
    mlp = fine.hip()
    t = torch.sort(2.0 + 4.0 * torch.rand(R, 192, device=dev), dim=-1).values
    o, d = batch[0], batch[1]
    n = R * 192
    rad, tape = T.forward_train(mlp, o, d, t)
    grad = torch.randn_like(rad)
    ms_fwd_infer = timed(lambda: mlp.eval_rays(o, d, t), 10)
    ms_fwd_tape = timed(lambda: T.forward_train(mlp, o, d, t), 10)
    import ctypes as C
    from nerfmeshes_amd import _lib
    from nerfmeshes_amd._lib import MlpDeltas, MlpTape
    lib = _lib.load()
    L, H = 8, 256
    f32 = dict(dtype=torch.float32, device=dev)
    dh, dfeat, dv, dlast = torch.empty(L, n, H, **f32), torch.empty(n, H, **f32), torch.empty(n, H // 2, **f32), torch.empty(n, 4, **f32)
    ct = MlpTape(*[C.c_void_p(tape[k].data_ptr()) for k in ("h", "feat", "v", "mask_h", "mask_v")])
    cd = MlpDeltas(*[C.c_void_p(x.data_ptr()) for x in (dh, dfeat, dv, dlast)])
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ms_bwd_kernel = timed(lambda: lib.nm_mlp_backward(mlp.handle, n, C.byref(ct), C.c_void_p(rad.data_ptr()),
                                                      C.c_void_p(grad.data_ptr()), C.byref(cd), stream), 10)
    ms_bwd_total = timed(lambda: T.backward(mlp, tape, rad, grad, o, d, t), 5)
    ms_gemm_one = timed(lambda: dh[3].t() @ tape["h"][2], 10)
    chunks = 96
    ms_gemm_split = timed(lambda: torch.bmm(dh[3].view(chunks, n // chunks, H).transpose(1, 2),
                                            tape["h"][2].view(chunks, n // chunks, H)).sum(0), 10)
    ms_refresh = timed(lambda: T.refresh(mlp, dict(fine.named_parameters())), 10)
    fwd_flops = mlp.flops_per_sample() * n
    bwd_macs = (128 * 256 + 8 * 256 * 256) * n
    res = {
        "config": f"8x256, 64+128 samples, {R} rays / iteration, perturb + noise, Adam",
        "ms_per_iteration": ms_iter, "rays_per_s": R / ms_iter * 1e3,
        "fine_net_192_samples": {
            "forward_inference_ms": ms_fwd_infer, "forward_taping_ms": ms_fwd_tape,
            "forward_taping_tflops": fwd_flops / ms_fwd_tape / 1e9,
            "backward_delta_kernel_ms": ms_bwd_kernel, "backward_delta_kernel_tflops": 2 * bwd_macs / ms_bwd_kernel / 1e9,
            "backward_total_ms (kernel + encodings + 14 weight-gradient GEMMs + bias sums)": ms_bwd_total,
            "one_weight_gradient_gemm_ms (256 x n x 256)": ms_gemm_one,
            "one_weight_gradient_gemm_split_k_bmm_ms": ms_gemm_split,
            "weight_gradient_gemm_tflops": 2 * 256 * 256 * n / ms_gemm_one / 1e9,
            "repack_ms": ms_refresh,
        },
    }
    # ---- the same iteration through torch autograd over the CPU oracle (bounded)
    if args.cpu_rays > 0:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
        from oracle import nerf_oracle as O
        spec = O.MLPSpec()
        rs = O.RenderSpec(training=True)
        Rc = args.cpu_rays
        oc, dc = o.cpu(), d[:Rc].cpu()
        tgt = target[:Rc].cpu()
        best = None
        for threads in (16, 32, 64):
            if threads > (os.cpu_count() or 1):
                continue
            torch.set_num_threads(threads)
            wc = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in model.model_coarse.named_parameters()}
            wf = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in model.model_fine.named_parameters()}

            def cpu_iter():
                t_c = O.perturb_intervals(O.coarse_intervals(2.0, 6.0, 64, Rc), torch.rand(Rc, 64))
                loss = 0.0
                tt = t_c
                for w_, first in ((wc, True), (wf, False)):
                    pts = O.ray_points(tt, dc, oc).reshape(-1, 3)
                    dd = dc[:, None, :].expand(-1, tt.shape[1], -1).reshape(-1, 3)
                    radc = O.mlp_forward(w_, spec, pts, dd, keep_graph=True).reshape(Rc, -1, 4)
                    b = O.composite(radc, tt, dc, rs, noise=0.2 * torch.randn(Rc, tt.shape[1]))
                    loss = loss + torch.nn.functional.mse_loss(b["rgb_map"], tgt)
                    if first:
                        tt = O.sample_pdf_intervals(t_c, b["weights"].detach(), 128, u=torch.rand(Rc, 128))
                loss.backward()

            cpu_iter()
            t0 = time.perf_counter()
            cpu_iter()
            dt = time.perf_counter() - t0
            if best is None or dt < best[0]:
                best = (dt, threads)
        res["cpu_oracle_autograd"] = {"rays": Rc, "seconds": best[0], "threads": best[1], "rays_per_s": Rc / best[0],
                                      "speedup": (R / ms_iter * 1e3) / (Rc / best[0])}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
