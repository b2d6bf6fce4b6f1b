"""Which torch ops (and from where) one training iteration launches besides the library's kernels:
python tests/tools/profile_train_ops.py [SHAPE]   (shape names of bench_train_shapes.py)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
import torch
from torch.profiler import profile, ProfilerActivity
import bench_train_shapes as B
name = sys.argv[1] if len(sys.argv) > 1 else "8x256"
iteration = B.build(name, torch.device("cuda:0"))[0]
for _ in range(3):
    iteration()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    iteration()
    torch.cuda.synchronize()
import collections
rows = collections.Counter()
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CPU and e.name.startswith("aten::") and e.cpu_parent is not None and not e.cpu_parent.name.startswith("aten::"):
        st = [s for s in (e.stack or []) if "nerfmeshes_amd" in s or "bench_train" in s or "optim" in s]
        rows[(e.name, st[0].split("/")[-1] if st else (e.cpu_parent.name[:40]))] += 1
    elif e.device_type == torch.autograd.DeviceType.CPU and e.name.startswith("aten::") and e.cpu_parent is None:
        st = [s for s in (e.stack or []) if "nerfmeshes_amd" in s or "bench_train" in s or "optim" in s]
        rows[(e.name, st[0].split("/")[-1] if st else "?")] += 1
for (n, w), c in sorted(rows.items(), key=lambda kv: -kv[1]):
    print(f"{c:4d}  {n:32s} {w}")
