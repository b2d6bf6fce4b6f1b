"""Every device copy / fill a training iteration issues besides the library's kernels: the aten ops behind them with the Python
frames that asked for them (the small networks' iteration is launch-bound: each is a 2 - 5 us kernel plus a launch gap).
    python tests/tools/profile_train_copies.py [SHAPE]"""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402
import bench_train_shapes as B  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "8x128"
iteration = B.build(name, torch.device("cuda:0"))[0]
for _ in range(3):
    iteration()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    iteration()
    torch.cuda.synchronize()
WATCH = ("aten::copy_", "aten::fill_", "aten::zero_", "aten::clone", "aten::_to_copy", "aten::zeros", "aten::zeros_like", "aten::ones_like",
         "aten::full", "aten::contiguous")
rows = collections.Counter()
for e in prof.events():
    if e.device_type != torch.autograd.DeviceType.CPU or e.name not in WATCH:
        continue
    parents, p = [], e.cpu_parent
    while p is not None and len(parents) < 4:
        parents.append(p.name[:40])
        p = p.cpu_parent
    frames = [s.split("/")[-1] for s in (e.stack or []) if "profiler" not in s and ("nerfmeshes_amd" in s or "bench_train" in s or "optim" in s)][:2]
    shape = getattr(e, "input_shapes", None)
    rows[(e.name, " < ".join(parents) or "-", " <- ".join(frames) or "?")] += 1
for (n, par, w), c in sorted(rows.items(), key=lambda kv: -kv[1])[:40]:
    print(f"{c:4d}  {n:16s} parents[{par}]  from[{w}]")
dev = collections.Counter(e.name[:60] for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA)
print("--- device-side events")
for n, c in dev.most_common(25):
    print(f"{c:4d}  {n}")
