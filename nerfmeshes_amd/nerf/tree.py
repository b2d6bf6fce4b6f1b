"""BuFF voxel tree (mirror of /root/reference/src/nerf/tree.py).

`Node` / `TreeSampling` keep the reference's names, attributes and checkpoint payload
(`serialize()` -> {"root", "voxels", "memm", "counter"}; the pickled `root` graph needs
`nerf.tree.Node` importable -- see nerfmeshes_amd.compat).  The inference hot spot,
`batch_ray_voxel_intersect` (tree.py:215-343, deterministic branch), runs as one gfx950 kernel with one
wavefront per ray and the voxel AABBs staged in LDS (nm_buff_intersect) instead of the reference's
dense (R, N, 3) temporaries.  Training-time tree maintenance: `ray_batch_integration`
(tree.py:177-206) accumulates on the GPU (nm_tree_integrate, two length-N accumulators instead of two dense
(R, N) scatter targets); `consolidate` (tree.py:127-175) is host-side bookkeeping over the node graph, as in the
reference.
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
        # Tie order of the voxel ids.  "reference": ties ordered as the reference's three unstable torch.sort calls order
        # them on the CPU -- its ids, hence its memm and its refined voxel sets, bit for bit
        # (tests/golden/buff_sampled_tree.npz).  "stable": every id is the voxel its sample lies in (6x faster, not the
        # reference's ids).  "auto" (default): "reference" while the owning model trains -- the ids are consumed by
        # ray_batch_integration only (model_buff.py:66-68) --, "stable" in eval, where nothing reads them.  The
        # optional hparams key `tree.tie_order` pins one of the three.
        self.tie_order = getattr(self.config.tree, "tie_order", "auto")
        self.training = False          # set by the owning model before each call (BuFFModel.forward)
        self.consolidate()

    def ticked(self, step):
        tree = self.config.tree
        if step > tree.step_size_integration_offset:
            cur = step - tree.step_size_integration_offset
            return cur > 0 and cur % tree.step_size_tree == 0
        return False

    def consolidate(self, split=False):
        """tree.py:127-175.  With accumulated weights: drop the voxels whose weight is <= tree.eps, then subdivide
        the survivors -- shallow nodes first, heavier first among equals -- for as long as the projected voxel count
        stays under tree.max_voxel_count; the rest are kept whole.  Always: rebuild the (N,2,3) voxel tensor, zero the
        weights, restart the integration counter."""
        if self.memm is not None:
            tree = self.config.tree
            from .. import dist as nd
            if nd.world()[1] > 1:
                # data-parallel training: every rank integrated its own rays; the refinement decision must be ONE
                # decision (the replicas' voxel sets may not drift apart), taken on the mean of the ranks' running means
                nd.all_reduce(self.memm)
                self.memm /= nd.world()[1]
            memm = self.memm.detach().cpu()
            print(f"Min memm {memm.min()}\nMax memm {memm.max()}\nMean memm {memm.mean()}\nMedian memm {memm.median()}")
            print(f"Threshold {tree.eps}")
            keep = torch.nonzero(memm > tree.eps).reshape(-1).tolist()
            lightness = (1.0 - memm[memm > tree.eps]).tolist()          # same order as `keep`
            print(f"From {memm.shape[0]} voxels with {memm.shape[0] - len(keep)} filtered to current {len(keep)}")
            survivors = [self.root.children[i] for i in keep]
            order = sorted(range(len(survivors)), key=lambda j: (survivors[j].depth, lightness[j]))   # stable
            growth = tree.subdivision_inner_count ** 3 - 1
            children = []
            for position, j in enumerate(order):
                node = survivors[j]
                projected = len(children) + growth + len(keep) - position
                if projected < tree.max_voxel_count:
                    node.subdivide()
                    children.extend(node.children if node.children else [node])
                else:
                    children.append(node)
            print(f"Now {len(children)} voxels")
            self.root.children = children
        voxels = [torch.stack(node.bounds, 0) for node in self.root.children]
        if not voxels:
            print(f"The chosen threshold {self.config.tree.eps} was set too high!")
        self.voxels = torch.stack(voxels, 0).to(self.device)          # (N, 2, 3) min / max corners
        self.memm = torch.zeros(self.voxels.shape[0]).to(self.device)
        self.counter = 1

    def ray_batch_integration(self, step, ray_voxel_indices, ray_batch_weights, ray_batch_weights_mask):
        """tree.py:177-206: running mean of the per-voxel sample weights of this ray batch (rays that hit the tree)."""
        offset = self.config.tree.step_size_integration_offset
        if step < offset:
            return
        if step == offset:
            print(f"Began ray batch integration... Step:{step}")
        hip_ops.tree_integrate(self.memm, self.counter, ray_voxel_indices, ray_batch_weights, ray_batch_weights_mask)
        self.counter += 1

    def batch_ray_voxel_intersect(self, origins, dirs, near, far, samples_count=64):
        """(z_vals (R,S) f32, voxel indices (R,S) i64, ray_mask (R,) bool) -- tree.py:215-343."""
        if self.config.tree.use_random_sampling:
            # tree.py:280-297: voxels drawn with torch.multinomial (one double per sample from the generator), depths
            # with torch.rand_like; here the draws come from torch's generator of the voxels' device
            dev = self.voxels.device
            shape = (dirs.reshape(-1, 3).shape[0], int(samples_count))
            u_pick = torch.rand(shape, dtype=torch.float64, device=dev)
            u_pos = torch.rand(shape, dtype=torch.float32, device=dev)
            return hip_ops.buff_intersect_random(self.voxels, origins, dirs, float(near), float(far), u_pick, u_pos)
        order = getattr(self, "tie_order", "auto")
        auto = order == "auto"
        if auto:
            # the reference's own ids (its three unstable sorts restated, NM_TIES_REFERENCE) in training AND in eval since round 5:
            # "bit-exact for index work" holds by default; 4.0 instead of 1.2 ms per 65 536 x 192 samples, < 3 % of a BuFF view.
            # `tree.tie_order = "stable"` selects the geometrically consistent order (every id = the voxel its sample lies in).
            order = "reference"
        if auto and order == "reference" and self.voxels.shape[0] > hip_ops.BUFF_REFERENCE_MAX_VOXELS:
            order = self._fall_back_to_stable(f"{self.voxels.shape[0]} voxels > {hip_ops.BUFF_REFERENCE_MAX_VOXELS}")
        return hip_ops.buff_intersect(self.voxels, origins, dirs, float(near), float(far), int(samples_count), ids=order)

    def _fall_back_to_stable(self, why):
        """`tie_order = "auto"` beyond the reference-order kernel's limit of 8192 voxels (its per-ray LDS holds the sort
        keys of every voxel; shipped configs: `tree.max_voxel_count` 1536): the stable order from here on, said once --
        "auto" never turns that limit into a failure in the middle of a training run; an explicit
        `tree.tie_order = "reference"` still raises.  (512 crossed voxels per ray is a limit of BOTH orders.)"""
        if not getattr(self, "_warned_tie_fallback", False):
            import warnings
            warnings.warn(f"tree.tie_order=auto: the reference tie order is not available ({why}); using the stable order "
                          "(the voxel ids then differ from the reference's, see DESIGN.md 3.4)")
            self._warned_tie_fallback = True
        self.tie_order = "stable"
        return "stable"

    def serialize(self):
        return {"root": self.root, "voxels": self.voxels, "memm": self.memm, "counter": self.counter}

    def deserialize(self, state):
        print("Loaded tree from checkpoint...")
        self.root = state["root"]
        self.voxels = state["voxels"].to(self.device)
        self.memm = state["memm"].to(self.device)
        self.counter = state["counter"]
