"""Fused-MLP kernel throughput for every instantiated network shape (nm_mlp_sample_points on 2^22 points):
algorithmic TFLOP/s (useful FLOP of the reference's layers, padding not counted) and the fraction of the fp32 MFMA peak."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from nerfmeshes_amd import hip_ops, synthetic as S

PEAK = 157.3
dev = torch.device("cuda:0")
n = 1 << 22
g = torch.Generator(device="cuda").manual_seed(0)
pts = (torch.rand(n, 3, device=dev, generator=g) * 2 - 1) * 2.0
dirs = torch.nn.functional.normalize(torch.randn(n, 3, device=dev, generator=g), dim=-1)
out = {}
for layers, hidden, fx, skip in ((8, 256, 10, 4), (8, 256, 6, 4), (8, 128, 10, 4), (8, 128, 6, 4), (6, 128, 6, 2),
                                 (8, 64, 10, 4), (4, 64, 6, 4), (9, 64, 10, 4)):
    kw = dict(num_layers=layers, hidden_size=hidden, skip_step=skip, num_encoding_fn_xyz=fx, num_encoding_fn_dir=4)
    mlp = hip_ops.HipMLP(S.make_mlp_weights(3, **kw), kw, dev)
    best = {}
    for name, fn in (("full", lambda: mlp.sample_points(pts, dirs)),):
        ts = []
        for _ in range(6):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        best[name] = min(ts[1:])
    tf = n * mlp.flops_per_sample() / (best["full"] * 1e-3) / 1e12
    key = f"{layers}x{hidden} F={fx}/4 skip {skip}"
    out[key] = {"ms": best["full"], "flops_per_sample": mlp.flops_per_sample(), "tflops": tf, "frac_of_fp32_mfma_peak": tf / PEAK,
                "points_per_s": n / (best["full"] * 1e-3)}
    print(f"{key:28s} {best['full']:8.3f} ms  {tf:6.1f} TFLOP/s  {tf / PEAK:.3f}  {n / best['full'] / 1e6:8.2f} G points/s")

# ---- generic-shape family (mlp_device_g.h): off-menu shapes, and the menu shapes forced onto it (what the tuning is worth)
def time_it(mlp):
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); mlp.sample_points(pts, dirs); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return min(ts[1:])


gen = {}
GENERIC = [dict(num_layers=8, hidden_size=512), dict(num_layers=8, hidden_size=256, num_encoding_fn_xyz=8),
           dict(num_layers=8, hidden_size=384), dict(num_layers=8, hidden_size=320), dict(num_layers=8, hidden_size=272),
           dict(num_layers=8, hidden_size=224), dict(num_layers=8, hidden_size=192), dict(num_layers=8, hidden_size=160),
           dict(num_layers=8, hidden_size=144), dict(num_layers=8, hidden_size=100), dict(num_layers=8, hidden_size=96),
           dict(num_layers=6, hidden_size=48, num_encoding_fn_xyz=6), dict(num_layers=4, hidden_size=32, num_encoding_fn_xyz=4, num_encoding_fn_dir=2, skip_step=2),
           dict(num_layers=8, hidden_size=256, num_encoding_fn_xyz=15, num_encoding_fn_dir=15),
           dict(num_layers=8, hidden_size=256, include_input_xyz=False, include_input_dir=False),
           dict(num_layers=8, hidden_size=400, use_viewdirs=False),
           # beyond the fused families: the layer-wise path (nerf_layerwise.hip; kernel_variant 2000)
           dict(num_layers=8, hidden_size=768), dict(num_layers=8, hidden_size=1024), dict(num_layers=8, hidden_size=256, num_encoding_fn_xyz=20)]
for over in GENERIC:
    kw = dict(dict(skip_step=4, num_encoding_fn_xyz=10, num_encoding_fn_dir=4, include_input_xyz=True, include_input_dir=True, use_viewdirs=True), **over)
    mlp = hip_ops.HipMLP(S.make_mlp_weights(3, **kw), kw, dev)
    variant, waves = mlp.kernel_variant()
    if variant < 1000:      # a tuned plan serves this shape (e.g. include_input_* off on a menu shape)
        variant = 1000
    ms = time_it(mlp)
    tf = n * mlp.flops_per_sample() / (ms * 1e-3) / 1e12
    key = " ".join(f"{k}={v}" for k, v in over.items())
    gen[key] = {"ms": ms, "flops_per_sample": mlp.flops_per_sample(), "tflops": tf, "frac_of_fp32_mfma_peak": tf / PEAK,
                "width_class_tiles": variant - 1000, "padded_hidden_size": 16 * (variant - 1000), "waves_per_workgroup": waves}
    print(f"generic {key:70s} class {variant - 1000:2d} {ms:8.3f} ms  {tf:6.1f} TFLOP/s  {tf / PEAK:.3f}")
for layers, hidden, fx, skip in ((8, 256, 10, 4), (8, 128, 10, 4), (4, 64, 6, 4)):
    kw = dict(num_layers=layers, hidden_size=hidden, skip_step=skip, num_encoding_fn_xyz=fx, num_encoding_fn_dir=4)
    mlp = hip_ops.HipMLP(S.make_mlp_weights(3, **kw), kw, dev, force_generic=True)
    ms = time_it(mlp)
    tf = n * mlp.flops_per_sample() / (ms * 1e-3) / 1e12
    key = f"{layers}x{hidden} F={fx}/4 skip {skip}"
    gen["menu shape on the generic family: " + key] = {"ms": ms, "tflops": tf, "frac_of_fp32_mfma_peak": tf / PEAK, "tuned_frac": out[key]["frac_of_fp32_mfma_peak"]}
    print(f"generic (forced) {key:28s} {ms:8.3f} ms  {tf:6.1f} TFLOP/s  {tf / PEAK:.3f}  (tuned {out[key]['frac_of_fp32_mfma_peak']:.3f})")
out["generic_family"] = gen
print(json.dumps(out))
