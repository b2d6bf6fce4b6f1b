"""One nm_weight_grad_batch launch in a loop (the PMC passes' target): JOBS products of OUT x IN over N rows whose operands are
ReLU-sparse like a real tape (half zeros).   python tests/tools/dw_batch_one.py OUT IN JOBS [N] [REPS]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from nerfmeshes_amd import hip_ops, synthetic as S, train_ops as T  # noqa: E402

o, i, jobs = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
n = int(sys.argv[4]) if len(sys.argv) > 4 else 2048 * 192
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
kw = dict(num_layers=4, hidden_size=128, skip_step=2, num_encoding_fn_xyz=6, num_encoding_fn_dir=4)
mlp = hip_ops.HipMLP({k: torch.as_tensor(v) for k, v in S.make_mlp_weights(3, **kw).items()}, kw, "cuda")
todo = []
for j in range(jobs):
    mask = torch.rand(n, o, device="cuda") < 0.5
    todo.append((torch.randn(n, o, device="cuda") * mask, torch.relu(torch.randn(n, i, device="cuda")), i, torch.empty(o, i, device="cuda"), 0,
                 torch.empty(o, device="cuda")))
for _ in range(reps):
    T._weight_grad_batch(mlp, todo)
torch.cuda.synchronize()
