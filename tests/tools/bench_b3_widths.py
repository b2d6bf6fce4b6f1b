"""The opt-in bf16x3 precision on the narrower shipped shapes (round 5: 8x128, 4x64 / 8x64) next to the fp32 kernels:
algorithmic (fp32-equivalent) TFLOP/s of nm_mlp_sample_points on 2^22 points."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from nerfmeshes_amd import hip_ops, synthetic as S
dev = torch.device("cuda:0"); n = 1 << 22
g = torch.Generator(device="cuda").manual_seed(0)
pts = (torch.rand(n, 3, device=dev, generator=g) * 2 - 1) * 2.0
dirs = torch.nn.functional.normalize(torch.randn(n, 3, device=dev, generator=g), dim=-1)
out = {}
for layers, hidden, fx in ((8, 256, 10), (8, 128, 10), (6, 128, 6), (8, 64, 10), (4, 64, 6)):
    kw = dict(num_layers=layers, hidden_size=hidden, skip_step=min(4, layers - 1), num_encoding_fn_xyz=fx, num_encoding_fn_dir=4)
    w = S.make_mlp_weights(3, density_gain=30.0, **kw)
    row = {}
    for prec in ("f32", "bf16x3"):
        mlp = hip_ops.HipMLP(w, kw, dev, precision=prec)
        ts = []
        for _ in range(6):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); res = mlp.sample_points(pts, dirs); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
        row[prec] = {"ms": min(ts[1:]), "tflops_fp32_equivalent": n * mlp.flops_per_sample() / (min(ts[1:]) * 1e-3) / 1e12}
        row[prec + "_out"] = res
    row["max_abs_drgb"] = float((row["f32_out"][:, :3] - row["bf16x3_out"][:, :3]).abs().max())
    row["speedup"] = row["f32"]["ms"] / row["bf16x3"]["ms"]
    del row["f32_out"], row["bf16x3_out"]
    out[f"{layers}x{hidden} F={fx}"] = row
    print(f"{layers}x{hidden} F={fx}", json.dumps(row), flush=True)
json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "gpurun_out", "b3_widths.json"), "w"), indent=1)
