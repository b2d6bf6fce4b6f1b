"""FlexibleNeRFModel with the reference's constructor, parameter names and forward signature
(/root/reference/src/nerf/models.py:4-80); forward runs the fused gfx950 MLP kernel."""
import torch

from .. import hip_ops, train_ops
from .modules import PositionalEncoding


class FlexibleNeRFModel(torch.nn.Module):
    def __init__(self, num_layers=4, hidden_size=128, skip_step=4, num_encoding_fn_xyz=6, num_encoding_fn_dir=4,
                 include_input_xyz=True, include_input_dir=True, log_sampling_xyz=True, log_sampling_dir=True,
                 use_viewdirs=True, **kwargs):
        super().__init__()
        self.encode_xyz = PositionalEncoding(num_encoding_fn_xyz, include_input_xyz, log_sampling_xyz)
        self.encode_dir = PositionalEncoding(num_encoding_fn_dir, include_input_dir, log_sampling_dir)
        self.dim_xyz = self.encode_xyz.output_size()
        self.dim_dir = self.encode_dir.output_size() if use_viewdirs else 0
        self.skip_step, self.num_layers, self.hidden_size, self.use_viewdirs = skip_step, num_layers, hidden_size, use_viewdirs
        lin = torch.nn.Linear
        self.layer1 = lin(self.dim_xyz, hidden_size)
        self.layers_xyz = torch.nn.ModuleList(
            lin(hidden_size + (self.dim_xyz if self._is_skip(i) else 0), hidden_size) for i in range(num_layers - 1))
        if use_viewdirs:
            self.layers_dir = torch.nn.ModuleList([lin(self.dim_dir + hidden_size, hidden_size // 2)])
            self.fc_alpha = lin(hidden_size, 1)
            self.fc_rgb = lin(hidden_size // 2, 3)
            self.fc_feat = lin(hidden_size, hidden_size)
        else:
            self.fc_out = lin(hidden_size, 4)
        self._desc = dict(num_layers=num_layers, hidden_size=hidden_size, skip_step=skip_step,
                          num_encoding_fn_xyz=num_encoding_fn_xyz, num_encoding_fn_dir=num_encoding_fn_dir,
                          include_input_xyz=include_input_xyz, include_input_dir=include_input_dir,
                          log_sampling_xyz=log_sampling_xyz, log_sampling_dir=log_sampling_dir,
                          use_viewdirs=use_viewdirs)
        self._hip = None
        self._hip_key = None
        self._pack = None         # [registration count, names, Parameter objects, storage pointers, nm_mlp_weights over them, owners]
        self._generation = 0      # advanced by train_ops' optimizer post-step hook when an optimizer holding these parameters steps
        self.weights_guard = None  # None: NERFMESHES_WEIGHTS_GUARD / "always"; or "always" | "key" | "check" for this module
        # arithmetic of the inference kernels: "f32" (default) or the opt-in "bf16x3" (hip_ops.HipMLP); training is fp32
        self.precision = "f32"

    def __getstate__(self):
        """copy.deepcopy (an EMA copy), pickling, torch.save(module): the device handle and the caches over it are run-time state
        of THIS object -- a copy builds its own on first use."""
        state = self.__dict__.copy()
        state["_hip"], state["_hip_key"], state["_pack"] = None, None, None
        return state

    def _is_skip(self, i):
        return i % self.skip_step == 0 and i > 0 and i != self.num_layers - 1

    def hip(self, precision=None):
        """Packed device copy of the current parameters: built once per device, then re-packed on the GPU (nm_mlp_refresh, one
        gather kernel on the stream, no host round trip) under the policy train_ops.guard_mode() names -- by default on EVERY
        use, so that the kernels see whatever the tensors hold now, as the reference's forward does
        (/root/reference/src/nerf/models.py:60-80 reads the parameters themselves); "key": only when the host can tell they
        changed; "check": "key" plus a device checksum that raises StaleWeightsError when the key missed an edit.
        `precision` overrides `self.precision` for this handle (mesh_nerf's density grid asks for "f32" whatever the
        module is set to)."""
        # the Parameter objects are cached (walking the module tree costs more than the re-pack it guards); torch tells us when
        # any module registers a parameter (train_ops.registrations()), which is the only way the list can change
        pack = getattr(self, "_pack", None)
        if pack is None or pack[0] != train_ops.registrations() or not train_ops.PARAMETER_HOOK or \
                any(owner._parameters.get(leaf) is not p for owner, leaf, p in pack[5]):      # (a direct `_parameters[...] = ` bypasses the hook)
            named = list(self.named_parameters())
            owners = []
            for n, p in named:
                path, _, leaf = n.rpartition(".")
                owners.append((self.get_submodule(path) if path else self, leaf, p))
            pack = self._pack = [train_ops.registrations(), [n for n, _ in named], [p for _, p in named], None, None, owners]
        names, params = pack[1], pack[2]
        dev = params[0].device
        if dev.type != "cuda":
            raise hip_ops._lib.HipLibraryError(
                "FlexibleNeRFModel lives on %s: move it to the MI355X (.to('cuda')); there is no CPU path" % dev)
        ptrs = tuple([p.data_ptr() for p in params])
        key = (train_ops.generation(), getattr(self, "_generation", 0), ptrs, tuple([p._version for p in params]))
        precision = "f32" if self.needs_grad() else (precision or getattr(self, "precision", "f32"))
        mode = train_ops.guard_mode(self)
        if self._hip is None or self._hip.device != dev or self._hip.precision != precision:
            self._hip = hip_ops.HipMLP(self.state_dict(), self._desc, dev, precision=precision)
            train_ops.register_owner(self, params)
            pack[3] = None
        elif mode == "always" or key != self._hip_key:
            if self._hip_key is None or ptrs != self._hip_key[2]:
                train_ops.register_owner(self, params)     # a parameter object / storage may have been replaced
            pack[4] = train_ops.refresh(self._hip, None, self._struct(pack, ptrs))
        elif mode == "check" and train_ops.weights_differ(self._hip, None, self._struct(pack, ptrs)):
            raise train_ops.StaleWeightsError(
                "the parameters of this FlexibleNeRFModel were edited in a way the host cannot see (p.data / torch._foreach_* on "
                ".data / a foreign kernel) and NERFMESHES_WEIGHTS_GUARD=check: the packed copy the HIP kernels read is stale.  "
                "Call .refresh() after such edits, or run under the default guard (\"always\": re-pack on every use)")
        self._hip_key = key
        return self._hip

    def _struct(self, pack, ptrs):
        """The nm_mlp_weights over the live storages, rebuilt only when a storage moved (26 ctypes fields otherwise per use)."""
        if pack[3] != ptrs or pack[4] is None:
            pack[4] = train_ops._weights_struct(self._hip, dict(zip(pack[1], pack[2])))
            pack[3] = ptrs
        return pack[4]

    def refresh(self):
        """Force a device re-pack on the next use (only the "key" / "check" guards ever need it: after edits through `p.data`)."""
        self._hip_key = None

    def refresh_count(self):
        """How many times the packed copy was (re)built since the handle exists (nm_mlp_refresh_count)."""
        return 0 if self._hip is None else self._hip.refresh_count()

    def needs_grad(self):
        return torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())

    def forward(self, ray_points, ray_directions=None):
        dirs = ray_directions if ray_directions is not None else ray_points
        if self.needs_grad():
            # differentiable path: every point is a one-sample ray (origin = point, t = 0: o + d * 0 == o exactly)
            pts = ray_points.reshape(-1, 3)
            t = torch.zeros(pts.shape[0], 1, dtype=torch.float32, device=pts.device)
            out = train_ops.mlp_rays(self, pts, dirs.expand_as(ray_points).reshape(-1, 3), t)
            return out.reshape(*ray_points.shape[:-1], 4)
        return self.hip().sample_points(ray_points, dirs)
