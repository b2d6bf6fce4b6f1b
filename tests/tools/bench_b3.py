"""Opt-in bf16x3 precision of the fused MLP: throughput and parity budget next to the default fp32 path.
Prints one JSON object (kept under profiles/)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from nerfmeshes_amd import hip_ops, synthetic as S
from oracle import nerf_oracle as O, parity
from tests.helpers import load_golden

dev = torch.device("cuda:0")
kw = dict(num_layers=8, hidden_size=256, skip_step=4, num_encoding_fn_xyz=10, num_encoding_fn_dir=4)
w = S.make_scene_weights(**kw)
f32 = hip_ops.HipMLP(w, kw, dev)
b3 = hip_ops.HipMLP(w, kw, dev, precision="bf16x3")
out = {}
# ---- throughput: 2^23 points through sample_points
n = 1 << 23
g = torch.Generator(device="cuda").manual_seed(0)
pts = (torch.rand(n, 3, device=dev, generator=g) * 2 - 1) * 2.0
dirs = torch.nn.functional.normalize(torch.randn(n, 3, device=dev, generator=g), dim=-1)
for name, m in (("f32", f32), ("bf16x3", b3)):
    m.sample_points(pts, dirs); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); r = m.sample_points(pts, dirs); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    out[name] = {"ms_2^23_points": min(ts), "algorithmic_tflops": n * m.flops_per_sample() / (min(ts) * 1e-3) / 1e12}
out["speedup"] = out["f32"]["ms_2^23_points"] / out["bf16x3"]["ms_2^23_points"]
# ---- per-sample error vs the CPU oracle (fp32) and vs an fp64 evaluation of the same network
m = 20000
po, do = pts[:m].cpu(), dirs[:m].cpu()
ref32 = O.mlp_forward(w, O.MLPSpec(**kw), po, do)
w64 = {k: torch.as_tensor(v).double() for k, v in w.items()}
ref64 = O.mlp_forward(w64, O.MLPSpec(**kw), po.double(), do.double(), keep_graph=True)
scale = float(ref64[:, 3].abs().max()) + 1.0
for name, mm in (("f32", f32), ("bf16x3", b3)):
    got = mm.sample_points(pts[:m], dirs[:m]).cpu().double()
    out[name].update({"max_abs_drgb_vs_fp64": float((got[:, :3] - ref64[:, :3]).abs().max()),
                      "max_abs_dsigma_over_scale_vs_fp64": float((got[:, 3] - ref64[:, 3]).abs().max()) / scale,
                      "rms_dsigma_over_scale_vs_fp64": float(((got[:, 3] - ref64[:, 3]) ** 2).mean().sqrt()) / scale})
out["oracle_fp32_vs_fp64"] = {"max_abs_drgb": float((ref32[:, :3].double() - ref64[:, :3]).abs().max()),
                              "max_abs_dsigma_over_scale": float((ref32[:, 3].double() - ref64[:, 3]).abs().max()) / scale}
# ---- render parity on the 8192-ray reference fixture + rays/s of a full 800x800 view
gfix = load_golden("render_lego_view_8k")
o, d = hip_ops.ray_bundle(gfix["pose"], 800, 800, S.LEGO_FOCAL_800, device=dev)
dd = d[torch.from_numpy(gfix["ray_index"]).to(dev)].contiguous()
near, far = torch.tensor([2.0]), torch.tensor([6.0])
uc, uf = torch.linspace(0, 1, 64), torch.linspace(0, 1, 128)
for name, mm in (("f32", f32), ("bf16x3", b3)):
    _, fb = hip_ops.render_rays(mm, mm, o[None], dd, near, far, uc, uf)
    out[name]["parity_view8k"] = parity.psnr_parity(fb["rgb_map"].cpu(), gfix["fine.rgb_map"], chunk=2048)
    hip_ops.render_rays(mm, mm, o[None], d[:65536], near, far, uc, uf); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for s in range(0, 640000, 65536):
        hip_ops.render_rays(mm, mm, o[None], d[s:s + 65536], near, far, uc, uf)
    b.record(); torch.cuda.synchronize()
    out[name]["rays_per_s_800x800_view"] = 640000 / (a.elapsed_time(b) * 1e-3)
# ---- mesh topology: density grid at res 160 with both precisions, marching cubes at the same iso
res = 160
ax = torch.linspace(-1.2, 1.2, res).to(dev)
g32 = f32.grid_query(ax, ax, ax, density_only=True).view(res, res, res)
g3 = b3.grid_query(ax, ax, ax, density_only=True).view(res, res, res)
iso = 32.0
m32 = hip_ops.marching_cubes(g32, iso); m3 = hip_ops.marching_cubes(g3, iso)
same_topology = m32[1].shape == m3[1].shape and bool(torch.equal(m32[1], m3[1]))
out["mesh_res160"] = {"vertices_f32": int(m32[0].shape[0]), "vertices_bf16x3": int(m3[0].shape[0]),
                      "faces_f32": int(m32[1].shape[0]), "faces_bf16x3": int(m3[1].shape[0]),
                      "identical_topology": same_topology,
                      "max_abs_dsigma_grid_over_scale": float((g32 - g3).abs().max() / (g32.abs().max() + 1)),
                      "sign_flips_at_iso": int(((g32 > iso) != (g3 > iso)).sum()),
                      "max_vertex_shift": float((m32[0] - m3[0]).abs().max()) if same_topology else None}
print(json.dumps(out))
