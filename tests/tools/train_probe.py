"""bench.py's training-iteration probe on its own (2048 rays, 8x256 coarse+fine, 64+128 samples)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from nerfmeshes_amd import hip_ops, synthetic as S
o, d = hip_ops.ray_bundle(S.orbit_poses(4)[0], 800, 800, S.LEGO_FOCAL_800, device="cuda")
print(json.dumps(bench.train_probe(torch.device("cuda"), d, o, iters=int(sys.argv[1]) if len(sys.argv) > 1 else 10)))
