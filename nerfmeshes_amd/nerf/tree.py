"""BuFF voxel tree (mirror of /root/reference/src/nerf/tree.py).

`Node` / `TreeSampling` keep the reference's names, attributes and checkpoint payload
(`serialize()` -> {"root", "voxels", "memm", "counter"}; the pickled `root` graph needs
`nerf.tree.Node` importable -- see nerfmeshes_amd.compat).  The inference hot spot,
`batch_ray_voxel_intersect` (tree.py:215-343, deterministic branch), runs as one gfx950 kernel with one
wavefront per ray and the voxel AABBs staged in LDS (nm_buff_intersect) instead of the reference's
dense (R, N, 3) temporaries.  Training-time tree maintenance (`ray_batch_integration`, the weighted
`consolidate`) is host-side bookkeeping outside this round's scope and raises.
"""
import torch

from .. import hip_ops


class Node:
    def __init__(self, config, bounds, depth):
        self.config = config
        self.bounds = bounds
        self.depth = depth
        self.max_depth = self.config.tree.max_depth
        self.count = (self.config.tree.subdivision_outer_count if depth == 0
                      else self.config.tree.subdivision_inner_count)
        self.weight = 0.0
        self.sparse = True
        self.children = []

    def subdivide(self):
        """Split into count^3 children, ordered x-major / z-fastest, child bounds computed as
        lo + idx / count * extent in fp32 (tree.py:19-33)."""
        if self.depth >= self.max_depth:
            return
        lo, hi = self.bounds
        extent = hi - lo
        n = self.count
        idx = torch.stack(torch.meshgrid(*(torch.arange(n, dtype=torch.float),) * 3, indexing="ij"), -1).view(-1, 3)
        for cell in idx:
            a = lo + cell / n * extent
            b = lo + (cell + 1.0) / n * extent
            self.children.append(Node(self.config, (a, b), self.depth + 1))

    def clear(self):
        self.children = []


class TreeSampling:
    def __init__(self, config, device):
        self.config = config
        self.device = device
        self.ray_near, self.ray_far = self.config.dataset.near, self.config.dataset.far
        self.ray_mean = (self.ray_near + self.ray_far) / 2
        bounds = (torch.tensor([self.ray_near - self.ray_mean] * 3), torch.tensor([self.ray_far - self.ray_mean] * 3))
        self.root = Node(self.config, bounds, 0)
        self.root.subdivide()
        self.voxels = None
        self.memm = None
        self.counter = 1
        self.consolidate()

    def ticked(self, step):
        tree = self.config.tree
        if step > tree.step_size_integration_offset:
            cur = step - tree.step_size_integration_offset
            return cur > 0 and cur % tree.step_size_tree == 0
        return False

    def consolidate(self, split=False):
        if self.memm is not None:
            raise NotImplementedError("weighted tree consolidation (BuFF training) is outside the HIP inference path")
        voxels = [torch.stack(node.bounds, 0) for node in self.root.children]
        self.voxels = torch.stack(voxels, 0).to(self.device)          # (N, 2, 3) min / max corners
        self.memm = torch.zeros(self.voxels.shape[0]).to(self.device)
        self.counter = 1

    def ray_batch_integration(self, step, ray_voxel_indices, ray_batch_weights, ray_batch_weights_mask):
        raise NotImplementedError("BuFF training-time weight integration is outside the HIP inference path")

    def batch_ray_voxel_intersect(self, origins, dirs, near, far, samples_count=64):
        """(z_vals (R,S) f32, voxel indices (R,S) i64, ray_mask (R,) bool) -- tree.py:215-343."""
        if self.config.tree.use_random_sampling:
            raise NotImplementedError("tree.use_random_sampling (multinomial branch) is not implemented on the HIP path")
        return hip_ops.buff_intersect(self.voxels, origins, dirs, float(near), float(far), int(samples_count))

    def serialize(self):
        return {"root": self.root, "voxels": self.voxels, "memm": self.memm, "counter": self.counter}

    def deserialize(self, state):
        print("Loaded tree from checkpoint...")
        self.root = state["root"]
        self.voxels = state["voxels"].to(self.device)
        self.memm = state["memm"].to(self.device)
        self.counter = state["counter"]
