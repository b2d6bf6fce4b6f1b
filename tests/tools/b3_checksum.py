"""SHA-256 of the bf16x3 kernel's outputs on fixed seeded inputs (points, density-only grid, one rendered view): run before
and after a change of mlp_device_b3.h that must not change a bit (the mode has no bit-exact oracle to compare with)."""
import hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from nerfmeshes_amd import hip_ops, synthetic as S

dev = torch.device("cuda:0")
out = {}
for fx in (10, 6):
    kw = dict(num_layers=8, hidden_size=256, skip_step=4, num_encoding_fn_xyz=fx, num_encoding_fn_dir=4)
    w = S.make_scene_weights(**kw) if fx == 10 else S.make_mlp_weights(5, density_gain=30.0, **kw)
    g = torch.Generator(device="cuda").manual_seed(3)
    n = (1 << 20) + 77
    pts = (torch.rand(n, 3, device=dev, generator=g) * 2 - 1) * 2.0
    dirs = torch.nn.functional.normalize(torch.randn(n, 3, device=dev, generator=g), dim=-1)
    h = lambda t: hashlib.sha256(t.contiguous().cpu().numpy().tobytes()).hexdigest()[:16]
    b3 = hip_ops.HipMLP(w, kw, dev, precision="bf16x3")
    out[f"fx{fx}"] = {"points": h(b3.sample_points(pts, dirs)),
                      "sigma": h(b3.grid_query(torch.linspace(-2, 2, 97), torch.linspace(-2, 2, 101), torch.linspace(-2, 2, 103)))}
print(json.dumps(out))
