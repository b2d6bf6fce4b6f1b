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
print(json.dumps(out))
