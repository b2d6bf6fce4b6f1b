"""A/B of the bf16x3 kernel against its two-column-tiles-per-wave experiment (mlp_device_b3w.h, ablation library,
NM_MLP_VARIANT=300) next to fp32: 2^23 points through sample_points, best of 7 over two rounds; one JSON object."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from nerfmeshes_amd import _lib, build as hip_build
if not os.path.exists(hip_build.ABLATION_LIB_PATH):
    hip_build.build(ablations=True, verbose=False)
_lib.LIB_PATH = hip_build.ABLATION_LIB_PATH            # explicit: nothing else in the package loads this library
from nerfmeshes_amd import hip_ops, synthetic as S

dev = torch.device("cuda:0")
kw = dict(num_layers=8, hidden_size=256, skip_step=4, num_encoding_fn_xyz=10, num_encoding_fn_dir=4)
w = S.make_scene_weights(**kw)
n = 1 << 23
g = torch.Generator(device="cuda").manual_seed(0)
pts = (torch.rand(n, 3, device=dev, generator=g) * 2 - 1) * 2.0
dirs = torch.nn.functional.normalize(torch.randn(n, 3, device=dev, generator=g), dim=-1)
out = {}
b3 = hip_ops.HipMLP(w, kw, dev, precision="bf16x3")
models = {"f32": hip_ops.HipMLP(w, kw, dev), "bf16x3_two_tiles": b3, "bf16x3_one_tile": b3}      # the variant is read at launch
res = {}
for rnd in range(2):
    for name, m in models.items():
        os.environ.pop("NM_MLP_VARIANT", None)
        if name == "bf16x3_two_tiles":
            os.environ["NM_MLP_VARIANT"] = "300"
        m.sample_points(pts, dirs); torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); res[name] = m.sample_points(pts, dirs); b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        best = min(ts + [out.get(name, {}).get("ms_2^23_points", 1e9)])
        out[name] = {"ms_2^23_points": best, "algorithmic_tflops": n * m.flops_per_sample() / (best * 1e-3) / 1e12}
out["two_tiles_equal_one_tile_bitwise"] = bool(torch.equal(res["bf16x3_two_tiles"], res["bf16x3_one_tile"]))
out["max_abs_diff_vs_f32"] = float((res["bf16x3_two_tiles"] - res["f32"]).abs().max())
print(json.dumps(out))
