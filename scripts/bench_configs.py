"""Secondary measurements through the drop-in class API (not the headline bench):
  * NeRFModel.query at the reference's validation chunksize (2048) and at 65536 rays per call,
  * BuFFModel.query (config 5 geometry: 504x378 rays, 192 samples, 12^3 voxel tree, bounds [0, 1.2]),
  * the tiny network of config 1 (4x64, 32 coarse samples, 400x400) on the GPU.
Prints one JSON object."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from nerfmeshes_amd import hip_ops, models, synthetic as S


def load(model, prefix, w):
    sd = model.state_dict()
    for k, v in w.items():
        sd[prefix + k] = torch.from_numpy(v)
    model.load_state_dict(sd)


def time_view(model, o, d, bounds, chunk, reps=2):
    def run():
        for s in range(0, d.shape[0], chunk):
            model.query((o, d[s:s + chunk], bounds))
    with torch.no_grad():
        run(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            run()
        torch.cuda.synchronize()
    return d.shape[0] * reps / (time.perf_counter() - t0)


out = {}
# --- NeRFModel through the class API
m = models.NeRFModel(S.hparams())
w = S.make_scene_weights()
load(m, "model_coarse.", w); load(m, "model_fine.", w)
m = m.eval().to("cuda")
o, d = hip_ops.ray_bundle(S.orbit_poses(4)[1], 800, 800, S.LEGO_FOCAL_800)
bounds = torch.tensor([2.0, 6.0])
for chunk in (2048, 65536):
    out[f"NeRFModel.query rays/s, 800x800 view, chunk {chunk}"] = time_view(m, o[None], d, bounds, chunk, reps=1)
# --- BuFF (config 5 geometry)
hp = S.hparams(model="BuFFModel", use_fine=False, num_coarse=192, num_fine=64, near=0.0, far=1.2, dataset_type="colmap")
b = models.BuFFModel(hp)
load(b, "model.", S.make_mlp_weights(9, density_gain=1500.0, density_bias=60.0))
b = b.eval().to("cuda")
H, W = 378, 504
pose = S.pose_spherical(30.0, -20.0, 1.0)
o2, d2 = hip_ops.ray_bundle(pose, H, W, 0.8 * W)
for chunk in (2048, 65536):
    out[f"BuFFModel.query rays/s, 504x378 view, 192 samples, chunk {chunk}"] = time_view(b, o2[None], d2, torch.tensor([0.0, 1.2]), chunk, reps=2)
with torch.no_grad():
    z, idx, mask = b.tree.batch_ray_voxel_intersect(o2[None], d2[:65536], 0.0, 1.2, 192)
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); b.tree.batch_ray_voxel_intersect(o2[None], d2[:65536], 0.0, 1.2, 192); e.record(); torch.cuda.synchronize()
out["nm_buff_intersect: 65536 rays x 1728 voxels x 192 samples, ms"] = a.elapsed_time(e)
out["BuFF rays hitting the tree"] = float(mask.float().mean())
# --- tiny (config 1 sizes) on the GPU
hp = S.hparams(hidden_size=64, num_layers=4, num_encoding_fn_xyz=6, num_coarse=32, num_fine=0, use_fine=False)
t = models.NeRFModel(hp)
load(t, "model_coarse.", S.make_mlp_weights(5, density_gain=100.0, hidden_size=64, num_layers=4, num_encoding_fn_xyz=6))
t = t.eval().to("cuda")
o3, d3 = hip_ops.ray_bundle(S.orbit_poses(4)[1], 400, 400, 555.5555)
out["tiny 4x64, 32 coarse: rays/s, 400x400 view, chunk 65536"] = time_view(t, o3[None], d3, bounds, 65536, reps=5)
print(json.dumps(out))
