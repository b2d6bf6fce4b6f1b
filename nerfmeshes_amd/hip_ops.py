"""Tensor-level host wrappers over the C ABI (torch is plumbing: device memory + streams).

Every function takes/returns torch CUDA(HIP) fp32 tensors, launches on torch's current stream and
raises if the HIP library is unavailable or an input is on the CPU -- no silent fallback.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import BundleOut, MlpDesc, MlpWeights, RenderCfg, check

BUNDLE_FIELDS = ("rgb_map", "depth_map", "weights", "mask_weights", "acc_map", "disp_map")


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(None)


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """torch's current stream on the current device as the `void* stream` of the C ABI (the raw accessor costs 0.3 us, building a
    torch.cuda.Stream object 5 -- 9: every wrapper below pays it once or twice per call)."""
    if _raw_stream is not None:
        return C.c_void_p(_raw_stream(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev32(t, device=None, name="tensor"):
    """contiguous fp32 tensor on the GPU; host tensors are accepted only where the reference
    hands over host data (ray bounds: eval_nerf.py:65, mesh_nerf.py:179)."""
    if not isinstance(t, torch.Tensor):
        t = torch.as_tensor(t, dtype=torch.float32)
    elif t.is_cuda and t.dtype is torch.float32 and t.is_contiguous() and (device is None or t.device == device):
        return t.detach()                      # the common case: nothing to convert
    if device is not None and t.device != device:
        t = t.to(device)
    if not t.is_cuda:
        raise _lib.HipLibraryError(f"{name} must live in GPU memory (got {t.device}); there is no CPU path")
    return t.detach().to(torch.float32).contiguous()


_small_cache = {}


def _dev32_small(t, device):
    """`_dev32` for the few-element HOST tensors that are the same call after call -- the ray bounds the reference keeps on
    the CPU (eval_nerf.py:65, mesh_nerf.py:179, every training batch): device copies cached by value.  A pageable H2D copy
    per call is a host-blocking hipMemcpy that first drains the stream: at the top of every training iteration / render
    chunk it left the GPU idle for the CPU's launch latency."""
    if isinstance(t, torch.Tensor) and not t.is_cuda and t.numel() <= 4:
        key = (str(device), tuple(float(x) for x in t.reshape(-1).tolist()))
        hit = _small_cache.get(key)
        if hit is None:
            if len(_small_cache) > 256:
                _small_cache.clear()
            hit = _small_cache[key] = t.detach().to(device=device, dtype=torch.float32).contiguous()
        return hit
    return _dev32(t, device)


PRECISIONS = {"f32": 0, "bf16x3": 1}


class HipMLP:
    """Device-resident packed copy of one FlexibleNeRFModel (handle of nm_mlp_create_ex)."""

    def __init__(self, state, desc, device, precision="f32", force_generic=False):
        """state: dict name -> array-like in torch.nn.Linear layout, keyed like
        FlexibleNeRFModel.state_dict(); desc: dict of constructor hyper-parameters.
        precision: "f32" (default: fp32 MFMA, the reference's arithmetic) or the opt-in "bf16x3" (every product emulated
        by six bf16 MFMA products of three-way operand splits, fp32 accumulation: fp32-class error, ~2x the throughput,
        inference only, the shipped 64- / 128- / 256-wide shapes).  force_generic: bind to the generic-shape kernel family even where a tuned
        kernel exists for the shape (NM_KERNEL_GENERIC: a cross-check, same results bit for bit)."""
        lib = _lib.load()
        if precision not in PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(PRECISIONS)}, got {precision!r}")
        self.precision = precision
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.HipLibraryError("HipMLP needs a GPU device")
        L = int(desc["num_layers"])
        host = {}

        def arr(key):
            v = state[key]
            a = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
            host[key] = np.ascontiguousarray(a, dtype=np.float32)
            return C.c_void_p(host[key].ctypes.data)

        fx, fd = int(desc["num_encoding_fn_xyz"]), int(desc["num_encoding_fn_dir"])
        for key, n, log in (("encode_xyz.frequency_bands", fx, desc.get("log_sampling_xyz", True)),
                            ("encode_dir.frequency_bands", fd, desc.get("log_sampling_dir", True))):
            if key not in state:
                state = dict(state)
                state[key] = (2.0 ** torch.linspace(0.0, n - 1, n)) if log else torch.linspace(1.0, 2.0 ** (n - 1), n)
        d = MlpDesc(L, int(desc["hidden_size"]), int(desc["skip_step"]), fx, fd,
                    int(bool(desc.get("include_input_xyz", True))), int(bool(desc.get("include_input_dir", True))),
                    int(bool(desc.get("use_viewdirs", True))))
        xs_w = (C.c_void_p * (L - 1))(*[arr(f"layers_xyz.{i}.weight") for i in range(L - 1)])
        xs_b = (C.c_void_p * (L - 1))(*[arr(f"layers_xyz.{i}.bias") for i in range(L - 1)])
        if d.use_viewdirs:
            w = MlpWeights(arr("layer1.weight"), arr("layer1.bias"), xs_w, xs_b,
                           arr("layers_dir.0.weight"), arr("layers_dir.0.bias"),
                           arr("fc_alpha.weight"), arr("fc_alpha.bias"), arr("fc_rgb.weight"), arr("fc_rgb.bias"),
                           arr("fc_feat.weight"), arr("fc_feat.bias"),
                           arr("encode_xyz.frequency_bands"), arr("encode_dir.frequency_bands"))
        else:
            # models.py:77-79: the trunk ends in fc_out (4, H): rows 0..2 are the colour rows, row 3 the density row
            if precision != "f32":
                raise _lib.HipLibraryError("use_viewdirs=False networks run in fp32 only")
            arr("fc_out.weight"), arr("fc_out.bias")
            H = int(desc["hidden_size"])
            ow, ob = host["fc_out.weight"].ctypes.data, host["fc_out.bias"].ctypes.data
            w = MlpWeights(arr("layer1.weight"), arr("layer1.bias"), xs_w, xs_b, C.c_void_p(None), C.c_void_p(None),
                           C.c_void_p(ow + 3 * H * 4), C.c_void_p(ob + 3 * 4), C.c_void_p(ow), C.c_void_p(ob),
                           C.c_void_p(None), C.c_void_p(None),
                           arr("encode_xyz.frequency_bands"), arr("encode_dir.frequency_bands"))
        self._h = C.c_void_p()
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        check(lib.nm_mlp_create_ex(C.byref(d), C.byref(w), idx, PRECISIONS[precision] | (0x100 if force_generic else 0),
                                   C.byref(self._h)), "nm_mlp_create_ex")
        self._lib = lib
        self.desc = dict(desc)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.nm_mlp_destroy(h)

    @property
    def handle(self):
        return self._h

    def refresh_count(self):
        return int(_lib.load().nm_mlp_refresh_count(self.handle))

    def kernel_variant(self):
        nw = C.c_int()
        return int(self._lib.nm_mlp_kernel_variant(self._h, C.byref(nw))), nw.value

    def flops_per_sample(self, density_only=False):
        return int(self._lib.nm_mlp_flops_per_sample(self._h, int(density_only)))

    def sample_points(self, points, dirs):
        points, dirs = _dev32(points, self.device, "points"), _dev32(dirs, self.device, "dirs")
        lead = points.shape[:-1]
        points, dirs = points.reshape(-1, 3), dirs.expand(*lead, 3).reshape(-1, 3).contiguous()
        out = torch.empty(points.shape[0], 4, dtype=torch.float32, device=self.device)
        check(self._lib.nm_mlp_sample_points(self._h, _ptr(points), _ptr(dirs), points.shape[0], _ptr(out), _stream()),
              "nm_mlp_sample_points")
        return out.reshape(*lead, 4)

    def eval_rays(self, origins, dirs, t):
        origins, dirs, t = (_dev32(x, self.device) for x in (origins, dirs, t))
        rays, samples = t.shape
        per_ray = int(origins.reshape(-1, 3).shape[0] == rays and rays > 1)
        out = torch.empty(rays, samples, 4, dtype=torch.float32, device=self.device)
        check(self._lib.nm_mlp_eval_rays(self._h, _ptr(origins), per_ray, _ptr(dirs), _ptr(t), rays, samples,
                                         _ptr(out), _stream()), "nm_mlp_eval_rays")
        return out

    def grid_query(self, ax0, ax1, ax2, first=0, count=None, density_only=True, out=None):
        ax = [_dev32(a, self.device) for a in (ax0, ax1, ax2)]
        n0, n1, n2 = (a.numel() for a in ax)
        count = n0 * n1 * n2 - first if count is None else count
        if out is None:
            out = torch.empty((count,) if density_only else (count, 4), dtype=torch.float32, device=self.device)
        check(self._lib.nm_mlp_grid_query(self._h, _ptr(ax[0]), _ptr(ax[1]), _ptr(ax[2]), n0, n1, n2, first, count,
                                          int(density_only), _ptr(out), _stream()), "nm_mlp_grid_query")
        return out


def ray_bundle(c2w, height, width, focal, first=0, count=None, device="cuda"):
    """get_ray_bundle on the GPU: (origin (3,) on `device`, dirs (count,3))."""
    lib = _lib.load()
    count = height * width - first if count is None else count
    pose = np.ascontiguousarray(np.asarray(c2w.detach().cpu() if isinstance(c2w, torch.Tensor) else c2w,
                                           dtype=np.float32)[:3, :4])
    dirs = torch.empty(count, 3, dtype=torch.float32, device=device)
    origin = np.zeros(3, dtype=np.float32)
    check(lib.nm_ray_bundle(pose.ctypes.data_as(_lib.c_float_p), height, width, float(focal), first, count,
                            _ptr(dirs), origin.ctypes.data_as(_lib.c_float_p), _stream()), "nm_ray_bundle")
    return torch.from_numpy(origin).to(device), dirs


def coarse_intervals(u, near, far, rays, lindisp=False):
    lib = _lib.load()
    u = _dev32(u)
    near, far = _dev32_small(near, u.device).reshape(-1), _dev32_small(far, u.device).reshape(-1)
    per_ray = int(near.numel() == rays and rays > 1)
    t = torch.empty(rays, u.numel(), dtype=torch.float32, device=u.device)
    check(lib.nm_coarse_intervals(_ptr(u), _ptr(near), _ptr(far), per_ray, int(lindisp), rays, u.numel(), _ptr(t),
                                  _stream()), "nm_coarse_intervals")
    return t


def _alloc_bundle(rays, samples, device, want=BUNDLE_FIELDS):
    shapes = dict(rgb_map=(rays, 3), depth_map=(rays,), weights=(rays, samples), mask_weights=(rays, samples),
                  acc_map=(rays,), disp_map=(rays,))
    tensors = {k: torch.empty(shapes[k], dtype=torch.float32, device=device) for k in want}
    out = BundleOut(*[_ptr(tensors.get(k)) for k in BUNDLE_FIELDS])
    return tensors, out


def composite(radiance, t, dirs, attenuation_threshold=1e-5, white_background=False, training=False):
    lib = _lib.load()
    radiance, t = _dev32(radiance), _dev32(t)
    dirs = _dev32(dirs, radiance.device)
    rays, samples = t.shape
    tensors, out = _alloc_bundle(rays, samples, radiance.device)
    check(lib.nm_composite(_ptr(radiance), _ptr(t), _ptr(dirs), rays, samples, float(attenuation_threshold),
                           int(white_background), int(training), C.byref(out), _stream()), "nm_composite")
    return tensors


def sample_pdf(t, weights, u):
    lib = _lib.load()
    t, weights = _dev32(t), _dev32(weights)
    u = _dev32(u, t.device)
    rays, coarse = t.shape
    out = torch.empty(rays, coarse + u.numel(), dtype=torch.float32, device=t.device)
    check(lib.nm_sample_pdf(_ptr(t), _ptr(weights), _ptr(u), rays, coarse, u.numel(), _ptr(out), _stream()),
          "nm_sample_pdf")
    return out


_workspaces = {}


def _workspace(nbytes, device):
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws


def render_rays(coarse, fine, origins, dirs, near, far, u_coarse, u_fine, lindisp=False, white_background=False,
                training=False, attenuation_threshold=1e-5):
    """NeRFModel.forward in one C call.  Returns (coarse dict, fine dict | None)."""
    lib = _lib.load()
    device = coarse.device
    origins, dirs = _dev32(origins, device, "origins").reshape(-1, 3), _dev32(dirs, device, "dirs").reshape(-1, 3)
    rays = dirs.shape[0]
    near, far = _dev32(near, device).reshape(-1), _dev32(far, device).reshape(-1)
    u_coarse = _dev32(u_coarse, device)
    sc = u_coarse.numel()
    nf = 0 if fine is None else int(u_fine.numel())
    u_f = None if fine is None else _dev32(u_fine, device)
    cfg = RenderCfg(sc, nf, int(lindisp), int(white_background), int(training), float(attenuation_threshold))
    ws = _workspace(int(lib.nm_render_workspace_bytes(rays, sc, nf)), device)
    ct, cout = _alloc_bundle(rays, sc, device)
    ft, fout = (None, None) if fine is None else _alloc_bundle(rays, sc + nf, device)
    per_ray_o = int(origins.shape[0] == rays and rays > 1)
    per_ray_b = int(near.numel() == rays and rays > 1)
    if rays == 0:      # an empty shard (more ranks than rays): empty bundles, no launch (zero-size tensors have no address to pass)
        return ct, ft
    check(lib.nm_render_rays(coarse.handle, fine.handle if fine is not None else None, C.byref(cfg), _ptr(origins),
                             per_ray_o, _ptr(dirs), _ptr(near), _ptr(far), per_ray_b, _ptr(u_coarse), _ptr(u_f),
                             rays, _ptr(ws), C.byref(cout), C.byref(fout) if fout is not None else None, _stream()),
          "nm_render_rays")
    return ct, ft


def make_view(c2w, height, width, focal, ndc_near=None):
    """nm_view from a (3|4, 4) camera-to-world matrix; ndc_near != None selects NDC rays (DataBundle.ndc uses 1.0)."""
    pose = np.ascontiguousarray(np.asarray(c2w.detach().cpu() if isinstance(c2w, torch.Tensor) else c2w,
                                           dtype=np.float32)[:3, :4]).reshape(-1)
    return _lib.View((C.c_float * 12)(*pose.tolist()), int(height), int(width), float(focal),
                     int(ndc_near is not None), float(ndc_near if ndc_near is not None else 1.0))


def view_rays(view, first=0, count=None, device="cuda"):
    """The rays nm_render_view generates, materialised: (origins (count,3), dirs (count,3))."""
    count = view.height * view.width - first if count is None else count
    o = torch.empty(count, 3, dtype=torch.float32, device=device)
    d = torch.empty(count, 3, dtype=torch.float32, device=device)
    check(_lib.load().nm_view_rays(C.byref(view), first, count, _ptr(o), _ptr(d), _stream()), "nm_view_rays")
    return o, d


def ndc_rays(height, width, focal, near, origins, dirs):
    """ndc_rays (nerf_helpers.py:280-307) on the GPU (nm_ndc_rays): origins (...,3) broadcastable to dirs (...,3)."""
    dirs = _dev32(dirs, name="rays_d")
    shape = dirs.shape
    d = dirs.reshape(-1, 3)
    origins = _dev32(origins, dirs.device, "rays_o")
    per_ray = origins.numel() != 3
    o = (origins.expand(shape).reshape(-1, 3) if per_ray else origins.reshape(1, 3)).contiguous()
    oo, od = torch.empty_like(d), torch.empty_like(d)
    check(_lib.load().nm_ndc_rays(int(height), int(width), float(focal), float(near), _ptr(o), int(per_ray), _ptr(d),
                                  d.shape[0], _ptr(oo), _ptr(od), _stream()), "nm_ndc_rays")
    return oo.reshape(shape), od.reshape(shape)


def positional_encoding(x, bands, include_input=True):
    """PositionalEncoding.forward (modules.py:26-34) on the GPU (nm_positional_encoding)."""
    x = _dev32(x, name="x")
    lead, dim = x.shape[:-1], x.shape[-1]
    b = np.ascontiguousarray(np.asarray(bands.detach().cpu() if isinstance(bands, torch.Tensor) else bands,
                                        dtype=np.float32))
    flat = x.reshape(-1, dim)
    out = torch.empty(flat.shape[0], 2 * dim * b.size + (dim if include_input else 0), dtype=torch.float32, device=x.device)
    check(_lib.load().nm_positional_encoding(_ptr(flat), flat.shape[0], dim, b.ctypes.data_as(_lib.c_float_p), b.size,
                                             int(bool(include_input)), _ptr(out), _stream()), "nm_positional_encoding")
    return out.reshape(*lead, out.shape[-1])


def render_view(coarse, fine, view, near, far, u_coarse, u_fine, first=0, count=None, lindisp=False,
                white_background=False, training=False, attenuation_threshold=1e-5):
    """NeRFModel.forward on rays generated in the kernels from the camera pose (nm_render_view): pixels
    [first, first+count) of `view` (hip_ops.make_view).  Returns (coarse dict, fine dict | None)."""
    lib = _lib.load()
    device = coarse.device
    count = view.height * view.width - first if count is None else count
    near, far = _dev32(near, device).reshape(-1), _dev32(far, device).reshape(-1)
    u_coarse = _dev32(u_coarse, device)
    sc = u_coarse.numel()
    nf = 0 if fine is None else int(u_fine.numel())
    u_f = None if fine is None else _dev32(u_fine, device)
    cfg = RenderCfg(sc, nf, int(lindisp), int(white_background), int(training), float(attenuation_threshold))
    ws = _workspace(int(lib.nm_render_workspace_bytes(count, sc, nf)), device)
    ct, cout = _alloc_bundle(count, sc, device)
    ft, fout = (None, None) if fine is None else _alloc_bundle(count, sc + nf, device)
    per_ray_b = int(near.numel() == count and count > 1)
    check(lib.nm_render_view(coarse.handle, fine.handle if fine is not None else None, C.byref(cfg), C.byref(view),
                             first, count, _ptr(near), _ptr(far), per_ray_b, _ptr(u_coarse), _ptr(u_f), _ptr(ws),
                             C.byref(cout), C.byref(fout) if fout is not None else None, _stream()), "nm_render_view")
    return ct, ft


def mlp_profile_enable(on=True):
    check(_lib.load().nm_mlp_profile_enable(int(on)), "nm_mlp_profile_enable")


def mlp_profile_read():
    """(launches, total kernel ms, total algorithmic flops) of the fused-MLP launches since the last read."""
    n, ms, fl = C.c_int64(), C.c_double(), C.c_double()
    check(_lib.load().nm_mlp_profile_read(C.byref(n), C.byref(ms), C.byref(fl)), "nm_mlp_profile_read")
    return n.value, ms.value, fl.value


TIE_ORDERS = {"stable": 0, "reference": 1}
BUFF_REFERENCE_MAX_VOXELS = 8192      # nm_buff_intersect_ex(NM_TIES_REFERENCE): sort keys of every voxel per ray in LDS


def buff_intersect(voxels, origins, dirs, near, far, samples, ids="stable"):
    """TreeSampling.batch_ray_voxel_intersect on the GPU (nm_buff_intersect_ex):
    (z (R,S) f32, voxel ids (R,S) i64, ray_mask (R,) bool).  ids="stable" (default): every id is the voxel its
    sample lies in; ids="reference": ties ordered as the reference's three unstable torch.sort calls order them on
    the CPU (libstdc++ introsort, evaluated wave-parallel in the kernel) -- the reference's ids bit for bit; what
    TreeSampling uses while training (tie_order="auto"), ~3x the stable order's time."""
    if ids not in TIE_ORDERS:
        raise ValueError(f"ids must be one of {sorted(TIE_ORDERS)}, got {ids!r}")
    lib = _lib.load()
    voxels = _dev32(voxels, name="voxels")
    dev = voxels.device
    origins, dirs = _dev32(origins, dev, "origins").reshape(-1, 3), _dev32(dirs, dev, "dirs").reshape(-1, 3)
    rays = dirs.shape[0]
    key = ("linspace", str(dev), int(samples))                    # tree.py:318; one H2D per (device, count), not per call
    u = _small_cache.get(key)
    if u is None:
        u = _small_cache[key] = torch.linspace(0, 1.0, samples).to(dev)
    z = torch.empty(rays, samples, dtype=torch.float32, device=dev)
    idx = torch.empty(rays, samples, dtype=torch.int64, device=dev)
    mask = torch.empty(rays, dtype=torch.uint8, device=dev)
    check(lib.nm_buff_intersect_ex(_ptr(voxels), voxels.shape[0], _ptr(origins),
                                   int(origins.shape[0] == rays and rays > 1), _ptr(dirs), float(near), float(far), _ptr(u),
                                   rays, samples, TIE_ORDERS[ids], _ptr(z), _ptr(idx), _ptr(mask), _stream()),
          "nm_buff_intersect_ex")
    return z, idx, mask.bool()


def buff_intersect_random(voxels, origins, dirs, near, far, u_pick, u_pos):
    """TreeSampling.batch_ray_voxel_intersect with `tree.use_random_sampling` (tree.py:280-297) on the GPU
    (nm_buff_intersect_random), as a function of the draws: u_pick (R,S) float64 -- what torch.multinomial consumes,
    one double per sample -- and u_pos (R,S) float32 (torch.rand_like).  Returns (z (R,S) f32 sorted, voxel ids (R,S)
    i64, ray_mask (R,) bool); rows of rays that cross no voxel are zero-filled (the caller overwrites them)."""
    lib = _lib.load()
    voxels = _dev32(voxels, name="voxels")
    dev = voxels.device
    origins, dirs = _dev32(origins, dev, "origins").reshape(-1, 3), _dev32(dirs, dev, "dirs").reshape(-1, 3)
    rays = dirs.shape[0]
    if not (isinstance(u_pick, torch.Tensor) and u_pick.is_cuda):
        raise _lib.HipLibraryError("u_pick must live in GPU memory; there is no CPU path")
    u_pick = u_pick.detach().to(device=dev, dtype=torch.float64).contiguous()
    u_pos = _dev32(u_pos, dev, "u_pos")
    if u_pick.dim() != 2 or u_pick.shape[0] != rays or tuple(u_pos.shape) != tuple(u_pick.shape):
        raise ValueError(f"u_pick / u_pos must both be (rays, samples) = ({rays}, S); got {tuple(u_pick.shape)}, {tuple(u_pos.shape)}")
    samples = u_pick.shape[1]
    z = torch.empty(rays, samples, dtype=torch.float32, device=dev)
    idx = torch.empty(rays, samples, dtype=torch.int64, device=dev)
    mask = torch.empty(rays, dtype=torch.uint8, device=dev)
    check(lib.nm_buff_intersect_random(_ptr(voxels), voxels.shape[0], _ptr(origins),
                                       int(origins.shape[0] == rays and rays > 1), _ptr(dirs), float(near), float(far),
                                       _ptr(u_pick), _ptr(u_pos), rays, samples, _ptr(z), _ptr(idx), _ptr(mask), _stream()),
          "nm_buff_intersect_random")
    return z, idx, mask.bool()


def tree_integrate(memm, counter, indices, weights, mask_weights):
    """TreeSampling.ray_batch_integration's arithmetic (nm_tree_integrate): updates `memm` (N,) in place from the
    (K,S) voxel ids / weights / visibility masks of the rays that hit the tree."""
    lib = _lib.load()
    memm = memm if (memm.is_cuda and memm.dtype == torch.float32 and memm.is_contiguous()) else None
    if memm is None:
        raise _lib.HipLibraryError("tree_integrate: memm must be a contiguous fp32 GPU tensor (updated in place)")
    dev = memm.device
    idx = indices.to(device=dev, dtype=torch.int64).contiguous()
    w, mw = _dev32(weights, dev, "weights"), _dev32(mask_weights, dev, "mask_weights")
    if not (idx.numel() == w.numel() == mw.numel()):
        raise ValueError("tree_integrate: indices / weights / mask_weights differ in size")
    ws = torch.empty(int(lib.nm_tree_workspace_bytes(memm.numel())), dtype=torch.uint8, device=dev)
    check(lib.nm_tree_integrate(_ptr(idx), _ptr(w), _ptr(mw), idx.numel(), memm.numel(), int(counter), _ptr(memm),
                                _ptr(ws), _stream()), "nm_tree_integrate")
    return memm


def np_stats(x):
    """numpy's fp32 statistics of a GPU tensor, bit for bit (nm_np_stats): dict(sum, mean, var, std, min, max) of python
    floats holding the fp32 values `x.cpu().numpy().sum() / .mean() / .var() / .std() / .min() / .max()` would give."""
    lib = _lib.load()
    x = _dev32(x, name="x").reshape(-1)
    if x.numel() == 0:
        raise ValueError("np_stats of an empty tensor")
    ws = torch.empty(int(lib.nm_np_stats_workspace_bytes(x.numel())), dtype=torch.uint8, device=x.device)
    out = np.zeros(6, dtype=np.float32)
    check(lib.nm_np_stats(_ptr(x), x.numel(), _ptr(ws), out.ctypes.data_as(_lib.c_float_p), _stream()), "nm_np_stats")
    return dict(zip(("sum", "mean", "var", "std", "min", "max"), (np.float32(v) for v in out)))


def np_stats_sharded(x_local, first, n_total, own_lo, own_hi, gather):
    """`np_stats` of an array of `n_total` elements that is spread over several ranks, numpy's fp32 result bit for bit
    (nm_np_chunk_sums / nm_np_finish).  `x_local` holds the global elements [first, first + x_local.numel()); this rank is
    responsible for the 8192-element chunks that START in [own_lo, own_hi) (their elements must be among the ones it
    holds: slabs carry a halo).  `gather(t)`: all-gather of a tensor with a ragged first dimension in rank order (identity
    on one rank); it is called once per pass (2 collectives in all).  Every rank returns the same dict."""
    lib = _lib.load()
    x = _dev32(x_local, name="x").reshape(-1)
    dev = x.device
    chunk = 8192
    chunks = int(lib.nm_np_chunk_count(int(n_total)))
    c_lo, c_hi = -(-int(own_lo) // chunk), min(-(-int(own_hi) // chunk), chunks)
    m = max(c_hi - c_lo, 0)
    out = np.zeros(6, dtype=np.float32)
    d_out = torch.empty(8, dtype=torch.float32, device=dev)
    mine = torch.empty(3, m + 1, dtype=torch.float32, device=dev)
    result = {}
    for second in (0, 1):
        if m:
            check(lib.nm_np_chunk_sums(_ptr(x), int(first), x.numel(), int(n_total), c_lo, c_hi, second,
                                       float(result.get("mean", 0.0)), _ptr(mine[0]), _ptr(mine[1]), _ptr(mine[2]), _stream()),
                  "nm_np_chunk_sums")
        # one gather per pass: the (sum, min, max) rows of a rank's chunks travel as one (m, 3) block
        nrows = 1 if second else 3
        packed = gather(mine[:nrows, :m].t().contiguous())
        if packed.shape[0] != chunks:
            raise RuntimeError(f"np_stats_sharded: the ranks cover {packed.shape[0]} of {chunks} chunks")
        rows = [packed[:, r].contiguous() for r in range(nrows)]
        check(lib.nm_np_finish(_ptr(rows[0]), _ptr(rows[1]) if not second else None, _ptr(rows[2]) if not second else None,
                               chunks, int(n_total), second, _ptr(d_out), out.ctypes.data_as(_lib.c_float_p), _stream()),
              "nm_np_finish")
        if not second:
            result.update(sum=np.float32(out[0]), mean=np.float32(out[1]), min=np.float32(out[4]), max=np.float32(out[5]))
        else:
            result.update(var=np.float32(out[2]), std=np.float32(out[3]))
    return result


def _mc_outputs(nv, nf, dev):
    """The four output arrays of a marching-cubes call as ONE allocation (256-byte aligned pieces): the host work between
    the count and the emit pass -- during which the GPU waits -- is two allocations and some integer arithmetic; the typed
    views are made after the kernels are launched.  Returns (buffer, device pointers, views())."""
    al = lambda b: (b + 255) & ~255
    sizes = (nv * 12, nf * 12, nv * 12, nv * 4)
    offs, total = [], 0
    for b in sizes:
        offs.append(total)
        total += al(b)
    out = torch.empty(max(total, 256), dtype=torch.uint8, device=dev)
    base = out.data_ptr()
    ptrs = tuple(C.c_void_p(base + o) for o in offs)

    def views():
        piece = lambda k, dt: out[offs[k]:offs[k] + sizes[k]].view(dt)
        return (piece(0, torch.float32).view(nv, 3), piece(1, torch.int32).view(nf, 3), piece(2, torch.float32).view(nv, 3),
                piece(3, torch.float32))

    return out, ptrs, views


def marching_cubes(volume, level):
    """skimage.measure.marching_cubes(volume, level) on the GPU (nm_mc_count + nm_mc_emit).
    volume: (n0,n1,n2) fp32 CUDA tensor.  Returns (verts (V,3) f32, faces (F,3) i32, normals (V,3) f32,
    values (V,) f32) as CUDA tensors; raises ValueError / RuntimeError exactly where skimage does."""
    lib = _lib.load()
    if not isinstance(volume, torch.Tensor) or volume.dim() != 3:
        raise ValueError("Input volume should be a 3D tensor.")
    if min(volume.shape) < 2:
        raise ValueError("Input array must be at least 2x2x2.")
    vol = _dev32(volume, name="volume")
    level = float(level)
    n0, n1, n2 = vol.shape
    dev = vol.device
    ws = torch.empty(int(lib.nm_mc_workspace_bytes(n0, n1, n2)), dtype=torch.uint8, device=dev)
    nv, nf = C.c_int64(), C.c_int64()
    p_vol, p_ws, stream = _ptr(vol), _ptr(ws), _stream()       # the GPU idles between the two calls: nothing is looked up twice
    check(lib.nm_mc_count(p_vol, n0, n1, n2, level, p_ws, C.byref(nv), C.byref(nf), stream), "nm_mc_count")
    if nv.value == 0:
        # skimage checks the level against the data range first (ValueError) and only then finds no surface
        # (RuntimeError).  A level outside [min, max] cannot produce a vertex, so the 442 MB min/max pass is only
        # paid on this error path instead of on every call.
        lo, hi = (float(v) for v in torch.aminmax(vol))
        if level < lo or level > hi:
            raise ValueError("Surface level must be within volume data range.")
        raise RuntimeError("No surface found at the given iso value.")
    out, (p_verts, p_faces, p_normals, p_values), views = _mc_outputs(nv.value, nf.value, dev)
    scratch = torch.empty(int(lib.nm_mc_vertex_scratch_bytes(nv.value, nf.value)) + 256, dtype=torch.uint8, device=dev)
    check(lib.nm_mc_emit(p_vol, n0, n1, n2, level, p_ws, _ptr(scratch), nv.value, nf.value, p_verts, p_faces, p_normals,
                         p_values, stream), "nm_mc_emit")
    verts, faces, normals, values = views()
    return verts, faces, normals, values


def marching_cubes_slab(volume, level, z_global, ghost_below, ghost_above):
    """Marching cubes of ONE axis-0 slab of a larger grid (nm_mc_count_slab / nm_mc_emit_slab).  `volume` holds the global
    planes [z_global, z_global + n0); its first cube layer is a ghost of the slab below (`ghost_below`), its last one a ghost
    of the slab above (`ghost_above`).  Returns an object with `.vertices` (the slab's own vertex count), `.faces`,
    `.ghost_vertices`, and `.emit(index_base)` -> (verts, faces, normals, values) with face entries = local id + index_base.
    Concatenating the ranks' arrays in rank order, with index_base = (own vertex counts of all lower ranks) - ghost_vertices,
    gives the mesh of the whole grid bit for bit (dist.marching_cubes_sharded does that)."""
    lib = _lib.load()
    if not isinstance(volume, torch.Tensor) or volume.dim() != 3:
        raise ValueError("Input volume should be a 3D tensor.")
    vol = _dev32(volume, name="volume")
    level = float(level)
    n0, n1, n2 = vol.shape
    dev = vol.device
    ws = torch.empty(int(lib.nm_mc_workspace_bytes(n0, n1, n2)), dtype=torch.uint8, device=dev)
    nv, nf, gv, gf = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
    check(lib.nm_mc_count_slab(_ptr(vol), n0, n1, n2, level, int(z_global), int(bool(ghost_below)), int(bool(ghost_above)),
                               _ptr(ws), C.byref(nv), C.byref(nf), C.byref(gv), C.byref(gf), _stream()), "nm_mc_count_slab")

    class Slab:
        vertices, faces = nv.value - gv.value, nf.value - gf.value
        ghost_vertices, ghost_faces = gv.value, gf.value

        @staticmethod
        def emit(index_base):
            verts = torch.empty(Slab.vertices, 3, dtype=torch.float32, device=dev)
            normals = torch.empty(Slab.vertices, 3, dtype=torch.float32, device=dev)
            values = torch.empty(Slab.vertices, dtype=torch.float32, device=dev)
            faces = torch.empty(Slab.faces, 3, dtype=torch.int32, device=dev)
            if nv.value:
                scratch = torch.empty(int(lib.nm_mc_vertex_scratch_bytes(nv.value, nf.value)) + 256, dtype=torch.uint8, device=dev)
                check(lib.nm_mc_emit_slab(_ptr(vol), n0, n1, n2, level, int(z_global), int(bool(ghost_below)), int(bool(ghost_above)),
                                          _ptr(ws), _ptr(scratch), nv.value, nf.value, gv.value, gf.value, int(index_base),
                                          _ptr(verts), _ptr(faces), _ptr(normals), _ptr(values), _stream()), "nm_mc_emit_slab")
            return verts, faces, normals, values

    return Slab

