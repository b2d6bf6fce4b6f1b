"""Every device copy / fill a training iteration issues besides the library's kernels, with the Python frames that asked for it
(the small networks' iteration is launch-bound: each of these is a 2 - 5 us kernel plus a launch gap).
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

name = sys.argv[1] if len(sys.argv) > 1 else "4x64 (config 1: 32 coarse, no fine)"
iteration = B.build(name, torch.device("cuda:0"))[0]
for _ in range(3):
    iteration()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    iteration()
    torch.cuda.synchronize()
WATCH = ("aten::copy_", "aten::fill_", "aten::zero_", "aten::clone", "aten::_to_copy", "aten::cat", "aten::index", "aten::mul", "aten::add",
         "aten::sub", "aten::div", "aten::sum", "aten::mean", "aten::randn", "aten::rand", "aten::randperm", "aten::sort", "aten::where",
         "aten::clamp", "aten::stack", "aten::expand", "aten::contiguous", "aten::zeros", "aten::ones", "aten::full", "aten::neg", "aten::sqrt")
rows = collections.Counter()
for e in prof.events():
    if e.device_type != torch.autograd.DeviceType.CPU or e.name not in WATCH:
        continue
    kernels = sum(1 for k in e.kernels) if hasattr(e, "kernels") else 0
    if not kernels:
        continue
    frames = [s.split("/")[-1] for s in (e.stack or []) if ("nerfmeshes_amd" in s or "bench_train" in s or "torch/optim" in s or "autograd" in s)
              and "profiler" not in s][:3]
    rows[(e.name, " <- ".join(frames) or "?")] += kernels
total = 0
for (n, w), c in sorted(rows.items(), key=lambda kv: -kv[1]):
    print(f"{c:4d}  {n:18s} {w}")
    total += c
print("device kernels from these ops:", total)
