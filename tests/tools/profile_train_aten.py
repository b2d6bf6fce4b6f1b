"""Which ATen operators (and how many device kernels each) one 8x256 training iteration launches besides the HIP library's own
kernels: torch.profiler over 3 iterations, grouped by operator name -- the tool that found the avoidable fills and copies of
DESIGN 3.5 (round 5).   python tests/tools/profile_train_aten.py"""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
import bench_train_shapes as B  # noqa: E402

dev = torch.device("cuda:0")
iteration = B.build("8x256", dev)[0]
for _ in range(3):
    iteration()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    for _ in range(3):
        iteration()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=45, max_name_column_width=70))
print(prof.key_averages(group_by_input_shape=True).table(sort_by="cuda_time_total", row_limit=60, max_name_column_width=50))
if "--stacks" in sys.argv:
    print(prof.key_averages(group_by_stack_n=6).table(sort_by="cuda_time_total", row_limit=60, max_name_column_width=50))
