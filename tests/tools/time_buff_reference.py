import torch, sys, os
sys.path.insert(0, "/root/repo")
from nerfmeshes_amd import hip_ops, synthetic as S
from oracle import nerf_oracle as O
dev = "cuda"
side = int(sys.argv[1]) if len(sys.argv) > 1 else 12
vox = O.buff_initial_voxels(0.0, 1.2, side).to(dev)
o, d = hip_ops.ray_bundle(S.pose_spherical(30.0, -20.0, 1.0), 378, 504, 0.8 * 504, device=dev)
d = d[:65536].contiguous()
def t(ids):
    hip_ops.buff_intersect(vox, o[None], d, 0.0, 1.2, 192, ids=ids)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(3): hip_ops.buff_intersect(vox, o[None], d, 0.0, 1.2, 192, ids=ids)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / 3
print("voxel grid %d^3, NM_R9_WAVES=%s: stable %.3f ms  reference %.3f ms" % (side, os.environ.get("NM_R9_WAVES"), t("stable"), t("reference")))
