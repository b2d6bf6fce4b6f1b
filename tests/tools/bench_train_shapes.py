"""Whole training iterations of networks OUTSIDE the tuned 8x256 / 8x128 shapes -- BASELINE config 1's 4x64, 8x64, generic-family
widths, a network without view directions, a ragged ray count -- through the module surface (forward in train mode: perturb +
noise, MSE(coarse) + MSE(fine), loss.backward(), Adam): ms per iteration and the iteration's algorithmic fp32 matrix work
(forward + delta propagation + weight gradients) over its WHOLE wall time against the fp32 MFMA peak.  With --trace-one NAME a
single shape runs a few iterations and exits (the target of `rocprofv3 --kernel-trace`: the trace must show no rocBLAS /
Cijk_* / at::native GEMM or reduction kernel between the forward and the optimizer).

    python tests/tools/bench_train_shapes.py [--iters 10] [--only 8x512,4x400] [--trace-one 8x320]
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if "--ablation-variant" in sys.argv:      # A/B against kernels only the ablation library holds (e.g. 310: the one-wave-per-SIMD wide classes)
    from nerfmeshes_amd import _lib, build as hip_build
    _lib.LIB_PATH = hip_build.ABLATION_LIB_PATH
    if sys.argv[sys.argv.index("--ablation-variant") + 1] != "0":
        os.environ["NM_MLP_VARIANT"] = sys.argv[sys.argv.index("--ablation-variant") + 1]
from nerfmeshes_amd import models, synthetic as S  # noqa: E402
from nerfmeshes_amd.nerf import CfgNode  # noqa: E402

PEAK = 157.3
SHAPES = {
    # name: (hparams overrides, rays, use_viewdirs)
    "4x64 (config 1: 32 coarse, no fine)": (dict(hidden_size=64, num_layers=4, skip_step=2, num_encoding_fn_xyz=6, num_coarse=32,
                                                 num_fine=0, use_fine=False), 8192, True),
    "4x64": (dict(hidden_size=64, num_layers=4, skip_step=2, num_encoding_fn_xyz=6), 2048, True),
    "8x64": (dict(hidden_size=64), 2048, True),
    "8x100": (dict(hidden_size=100), 2048, True),
    "8x128": (dict(hidden_size=128), 2048, True),
    "8x160": (dict(hidden_size=160), 2048, True),
    "8x256": (dict(), 2048, True),
    "8x256 ragged (583 rays)": (dict(), 583, True),
    "8x320": (dict(hidden_size=320), 2048, True),
    "4x400 flat": (dict(hidden_size=400, num_layers=4, skip_step=2), 2048, False),
    "8x448": (dict(hidden_size=448), 1024, True),
    "8x512": (dict(hidden_size=512), 1024, True),
    "8x768 (layer-wise path)": (dict(hidden_size=768), 512, True),
    "8x1024 (layer-wise path)": (dict(hidden_size=1024), 512, True),
}


def flops_per_sample(kw, viewdirs):
    """(forward, delta, weight-gradient) algorithmic FLOP per sample of FlexibleNeRFModel (weights only, as SURVEY 8(d))."""
    H, L, ss = kw["hidden_size"], kw["num_layers"], kw["skip_step"]
    dx, dd = 6 * kw["num_encoding_fn_xyz"] + 3, 6 * kw["num_encoding_fn_dir"] + 3
    nskip = sum(1 for i in range(L - 1) if i % ss == 0 and i > 0 and i != L - 1)
    trunk = dx * H + (L - 1) * H * H + nskip * dx * H
    if viewdirs:
        fwd = trunk + H * H + H + (H + dd) * (H // 2) + 3 * (H // 2)
        delta = (L - 1) * H * H + H * H + H * (H // 2)
    else:
        fwd = trunk + 4 * H
        delta = (L - 1) * H * H
    return 2 * fwd, 2 * delta, 2 * fwd


def build(name, dev):
    over, rays, viewdirs = SHAPES[name]
    kw = dict(hidden_size=256, num_layers=8, skip_step=4, num_encoding_fn_xyz=10, num_encoding_fn_dir=4, num_coarse=64, num_fine=128,
              use_fine=True)
    kw.update(over)
    hp = S.hparams(train_perturb=True, train_noise_std=0.2, **kw)
    if not viewdirs:
        for part in ("coarse", "fine"):
            hp[f"models.{part}.use_viewdirs"] = False
        hp["nerf.use_viewdirs"] = False
    torch.manual_seed(0)
    model = models.NeRFModel(CfgNode(hp)).to(dev)
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=5e-4)
    g = torch.Generator().manual_seed(1)
    dirs = torch.nn.functional.normalize(torch.randn(rays, 3, generator=g), dim=-1).to(dev)
    batch = (torch.tensor([[0.0, 0.0, 4.0]], device=dev), dirs, torch.tensor([2.0, 6.0]))
    target = torch.rand(rays, 3, generator=g).to(dev)

    def iteration():
        opt.zero_grad(set_to_none=True)
        out = model(batch)
        c, f = out if isinstance(out, tuple) else (out, None)
        loss = torch.nn.functional.mse_loss(c.rgb_map, target)
        if f is not None:
            loss = loss + torch.nn.functional.mse_loss(f.rgb_map, target)
        loss.backward()
        opt.step()

    samples = rays * (kw["num_coarse"] + (kw["num_coarse"] + kw["num_fine"] if kw["use_fine"] else 0))
    return iteration, kw, viewdirs, rays, samples, model


def main():
    dev = torch.device("cuda:0")
    iters = int(sys.argv[sys.argv.index("--iters") + 1]) if "--iters" in sys.argv else 10
    if "--trace-one" in sys.argv:
        name = sys.argv[sys.argv.index("--trace-one") + 1]
        iteration = build(name, dev)[0]
        for _ in range(3):
            iteration()
        torch.cuda.synchronize()
        return
    out = {}
    only = sys.argv[sys.argv.index("--only") + 1].split(",") if "--only" in sys.argv else None
    for name in SHAPES:
        if only is not None and not any(name.startswith(o) for o in only):
            continue
        iteration, kw, viewdirs, rays, samples, model = build(name, dev)
        for _ in range(3):
            iteration()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            iteration()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / iters * 1e3
        from nerfmeshes_amd import train_ops
        per_iter = []
        for _ in range(iters):                 # medians over the iterations: one allocator / launch hiccup must not skew a stage
            train_ops.profile_stages(True)
            iteration()
            per_iter.append(train_ops.profile_stages(False))
        stages = {k: round(sorted(d.get(k, 0.0) for d in per_iter)[len(per_iter) // 2], 3) for k in per_iter[0]}
        stages["rest"] = round(ms - sum(stages.values()), 3)
        fwd, delta, dw = flops_per_sample(kw, viewdirs)
        variant = model.model_coarse.hip().kernel_variant()[0]
        if variant < 1000 and kw["num_layers"] >= 3:
            # layer1's gradient by linearity (tuned family: train_ops.backward / the fused backward): the executed delta chain leaves
            # layers_xyz[0]^T out for the networks whose sample count takes that path -- the FLOP counted here are the executed ones
            from benchlib.train import _linear_layer1_share
            per_net = (kw["num_coarse"],) + ((kw["num_coarse"] + kw["num_fine"],) if kw["use_fine"] else ())
            share = _linear_layer1_share(kw["hidden_size"], rays, per_net)
            delta -= 2 * kw["hidden_size"] ** 2 * share            # layers_xyz[0]^T not applied per sample
            dw -= 2 * kw["hidden_size"] ** 2 * share               # layers_xyz[0]'s own weight gradient from the sums
        flops = samples * (fwd + delta + dw)
        out[name] = {"rays": rays, "samples_per_iteration": samples, "ms_per_iteration": round(ms, 3),
                     "rays_per_s": round(rays / ms * 1e3), "kernel_family": "layer-wise" if variant == 2000 else ("generic class %d" % (variant - 1000) if variant >= 1000 else "tuned"),
                     "algorithmic_tflop_per_iteration": round(flops / 1e12, 4),
                     "floor_ms_at_fp32_mfma_peak": round(flops / (PEAK * 1e12) * 1e3, 3),
                     "frac_of_fp32_mfma_peak_whole_iteration": round(flops / (ms * 1e-3) / 1e12 / PEAK, 3),
                     "stage_ms": stages,
                     "stage_frac": {"taping_forward": round(samples * fwd / (stages.get("taping_forward", 1e9) * 1e-3) / 1e12 / PEAK, 3),
                                    "delta": round(samples * delta / (stages.get("delta", 1e9) * 1e-3) / 1e12 / PEAK, 3),
                                    "weight_gradients": round(samples * dw / (stages.get("weight_gradients", 1e9) * 1e-3) / 1e12 / PEAK, 3)}}
        print(name, json.dumps(out[name]), flush=True)
        del iteration, model
        torch.cuda.empty_cache()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    tag = ("_variant_" + sys.argv[sys.argv.index("--ablation-variant") + 1]) if "--ablation-variant" in sys.argv else ""
    json.dump({"peak_tflops": PEAK, "iters": iters, "shapes": out}, open(os.path.join(ROOT, "gpurun_out", f"train_shapes{tag}.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
