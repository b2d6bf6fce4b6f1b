"""Where the host time of the unmodified mesh_nerf.py loop goes over the HIP path (1024-point sample_points calls + .cpu()):
cProfile of benchlib.mesh.reference_batch_probe's inner loop.   python tests/tools/profile_route_a.py [--guard key]"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if "--guard" in sys.argv:
    os.environ["NERFMESHES_WEIGHTS_GUARD"] = sys.argv[sys.argv.index("--guard") + 1]
import torch  # noqa: E402
from nerfmeshes_amd import models, synthetic as S  # noqa: E402
from nerfmeshes_amd.nerf.nerf_helpers import batchify  # noqa: E402

dev = torch.device("cuda:0")
model = models.NeRFModel(S.hparams()).eval().to(dev)
pts = torch.rand(1024 * 2000, 3) * 2.4 - 1.2


def run(points):
    got = []
    with torch.no_grad():
        for (x,) in batchify(points, batch_size=1024, device=dev, progress=False):
            got.append(model.sample_points(x, x).cpu())
    return got


run(pts[:8192])
torch.cuda.synchronize()
t0 = time.perf_counter()
run(pts)
dt = time.perf_counter() - t0
print(f"{dt / 2000 * 1e6:.1f} us per call")
pr = cProfile.Profile()
pr.enable()
run(pts)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
