import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from nerfmeshes_amd import models, synthetic as S, train_ops as T, hip_ops
from nerfmeshes_amd.nerf import CfgNode
dev = torch.device("cuda:0")
model = models.NeRFModel(CfgNode(S.hparams(train_perturb=True, train_noise_std=0.2))).to(dev)
model.train()
mlp = model.model_fine.hip()
fl = mlp.flops_per_sample() if hasattr(mlp, "flops_per_sample") else 1.19e6
R = 2048
origin, dirs = hip_ops.ray_bundle(S.pose_spherical(30.0, -30.0, 4.0), 800, 800, 1111.1111, 0, 640000, dev)
d = dirs[:R].contiguous(); o = origin[None]
def timed(fn, it=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it
for s in (64, 128, 192, 384, 768):
    t = torch.sort(2.0 + 4.0 * torch.rand(R, s, device=dev), dim=-1).values
    ms_t = timed(lambda: T.forward_train(mlp, o, d, t))
    ms_i = timed(lambda: mlp.eval_rays(o, d, t))
    rad, tape = T.forward_train(mlp, o, d, t)
    g = torch.randn_like(rad)
    ms_b = timed(lambda: T.backward(mlp, tape, rad, g, o, d, t))
    n = R * s
    print(f"samples {s:4d} n={n:8d}  tape fwd {ms_t:.3f} ms ({n*fl/ms_t/1e9:.1f} TF)  infer {ms_i:.3f} ms ({n*fl/ms_i/1e9:.1f} TF)  backward(all) {ms_b:.3f} ms")

# the same two calls interleaved, as a training iteration runs them: each leg timed with its own event pair
t = torch.sort(2.0 + 4.0 * torch.rand(R, 192, device=dev), dim=-1).values
ev = lambda: torch.cuda.Event(enable_timing=True)
acc = [0.0, 0.0]
for it in range(13):
    e = [ev() for _ in range(3)]
    e[0].record()
    rad, tape = T.forward_train(mlp, o, d, t)
    e[1].record()
    T.backward(mlp, tape, rad, g_ := torch.ones_like(rad), o, d, t)
    e[2].record()
    torch.cuda.synchronize()
    if it >= 3:
        acc[0] += e[0].elapsed_time(e[1]); acc[1] += e[1].elapsed_time(e[2])
print(f"interleaved (192 samples): tape fwd {acc[0]/10:.3f} ms  backward(all) {acc[1]/10:.3f} ms")
import subprocess
print(subprocess.run("rocm-smi --showclocks --showpower 2>/dev/null | grep -i 'sclk\\|power' | head -4", shell=True, capture_output=True, text=True).stdout)
