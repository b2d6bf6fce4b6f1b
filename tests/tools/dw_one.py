"""One weight-gradient product in a loop (profiling target): python tests/tools/dw_one.py OUT IN [N] [REPS]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from nerfmeshes_amd import _lib
o, i = int(sys.argv[1]), int(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 2048 * 192
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
lib = _lib.load(); dev = torch.device("cuda:0")
cus = torch.cuda.get_device_properties(dev).multi_processor_count
d = torch.randn(n, o, device=dev); a = torch.randn(n, i, device=dev)
ws = torch.empty(max(int(lib.nm_weight_grad_workspace_bytes_ex(o, o, i, i, cus)), 1 << 28), dtype=torch.uint8, device=dev)
dw, db = torch.empty(o, i, device=dev), torch.empty(o, device=dev)
p = lambda x: C.c_void_p(x.data_ptr())
for _ in range(reps):
    assert lib.nm_weight_grad_ex(cus, p(d), o, o, p(a), i, i, n, p(ws), p(dw), i, 0, p(db), None) == 0
torch.cuda.synchronize()
