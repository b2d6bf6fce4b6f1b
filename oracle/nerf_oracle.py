"""CPU ORACLE (test infrastructure, NOT product code) for the NeRF ray-batch path.

A functional, stateless restatement -- in torch *CPU* fp32 ops, issued in the
same order as the reference so that it is bit-identical to it on the same torch
build -- of the reference hot path:

  R0  get_ray_bundle / ndc_rays          /root/reference/src/nerf/nerf_helpers.py:226-307
  R1  RaySampleInterval.forward          /root/reference/src/nerf/modules.py:157-186
  R2  intervals_to_ray_points            /root/reference/src/models/model_helpers.py:32-35
  R3a PositionalEncoding.forward         /root/reference/src/nerf/modules.py:26-34
  R3b FlexibleNeRFModel.forward          /root/reference/src/nerf/models.py:60-80
  R4  VolumeRenderer.forward             /root/reference/src/nerf/modules.py:67-121
      cumprod_exclusive                  /root/reference/src/nerf/nerf_helpers.py:199-223
  R5  SamplePDF.forward / sample_pdf     /root/reference/src/nerf/modules.py:197-248
  R6  NeRFModel.forward / query          /root/reference/src/models/model_nerf.py:37-86
  R7  BaseModel.sample_points            /root/reference/src/models/model_base.py:65-73
  R8  eval_nerf loss/PSNR bookkeeping    /root/reference/src/eval_nerf.py:50-105
  R11 extract_radiance / iso level       /root/reference/src/mesh_nerf.py:27-92

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may
import this module, and only as the checker / timed CPU baseline.  The product
(`nerfmeshes_amd`) never imports it and has no CPU fallback.

Parity pin: the reference has no tests or golden vectors of its own
(SURVEY.md F2).  This oracle is pinned instead against outputs of the
unmodified reference executed in the build container
(`tests/golden/make_golden.py` -> `tests/golden/*.npz`, checked by
`tests/test_oracle_golden.py`) -- bit-exact there, and within 1e-5 on hosts
whose BLAS picks a different summation order.

Weights are passed as a plain dict keyed exactly like the reference
`FlexibleNeRFModel.state_dict()` (`layer1.weight`, `layers_xyz.3.bias`, ...).
"""
from dataclasses import dataclass

import numpy as np
import torch


@dataclass(frozen=True)
class MLPSpec:
    """Hyper-parameters of FlexibleNeRFModel (models.py:5-19)."""
    num_layers: int = 8
    hidden_size: int = 256
    skip_step: int = 4
    num_encoding_fn_xyz: int = 10
    num_encoding_fn_dir: int = 4
    include_input_xyz: bool = True
    include_input_dir: bool = True
    log_sampling_xyz: bool = True
    log_sampling_dir: bool = True
    use_viewdirs: bool = True

    @property
    def dim_xyz(self):
        return 6 * self.num_encoding_fn_xyz + (3 if self.include_input_xyz else 0)

    @property
    def dim_dir(self):
        if not self.use_viewdirs:
            return 0
        return 6 * self.num_encoding_fn_dir + (3 if self.include_input_dir else 0)

    def is_skip(self, i):
        # models.py:37,63 -- layer i of layers_xyz consumes cat(hidden, xyz_enc)
        return i % self.skip_step == 0 and i > 0 and i != self.num_layers - 1


@dataclass(frozen=True)
class RenderSpec:
    """The subset of cfg.nerf.{train,validation} + cfg.dataset the path reads."""
    num_coarse: int = 64
    num_fine: int = 128
    lindisp: bool = False
    perturb: bool = False          # only the deterministic branch is restated
    white_background: bool = False
    attenuation_threshold: float = 1e-5   # model_base.py:32
    noise_std: float = 0.0
    training: bool = False


def _t(x):
    if isinstance(x, torch.Tensor):
        return x.detach().to(torch.float32).cpu()
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))


def frequency_bands(n, log_sampling=True):
    """modules.py:16-23"""
    if log_sampling:
        return 2.0 ** torch.linspace(0.0, n - 1, n)
    return torch.linspace(2.0 ** 0.0, 2.0 ** (n - 1), n)


def positional_encoding(x, n_freq, include_input=True, log_sampling=True):
    """R3a.  Coordinate-major layout: [x | sin(x_c * f_k) for c, k | cos(...)]."""
    bands = frequency_bands(n_freq, log_sampling)
    shape = list(x.shape)
    scaled = (bands * x[..., None].expand(*shape, n_freq)).reshape(*shape[:-1], -1)
    parts = ([x] if include_input else []) + [torch.sin(scaled), torch.cos(scaled)]
    return torch.cat(parts, dim=-1)


def mlp_forward(w, spec: MLPSpec, points, directions=None, keep_graph=False):
    """R3b.  (N,3),(N,3) -> (N,4) = [sigmoid rgb | raw sigma].  keep_graph: use the tensors as given (any
    dtype, autograd graph intact) -- the gradient checks of the training path differentiate through this."""
    lin = torch.nn.functional.linear
    conv = (lambda x: x) if keep_graph else _t
    w = {k: conv(v) for k, v in w.items()}
    points = conv(points)
    enc = positional_encoding(points, spec.num_encoding_fn_xyz, spec.include_input_xyz, spec.log_sampling_xyz)
    h = lin(enc, w["layer1.weight"], w["layer1.bias"])          # no activation (models.py:62)
    for i in range(spec.num_layers - 1):
        if spec.is_skip(i):
            h = torch.cat((h, enc), dim=-1)                      # hidden first (models.py:65)
        h = torch.relu(lin(h, w[f"layers_xyz.{i}.weight"], w[f"layers_xyz.{i}.bias"]))
    if not spec.use_viewdirs:
        out = lin(h, w["fc_out.weight"], w["fc_out.bias"])
        out[..., :3] = torch.sigmoid(out[..., :3])
        return out
    view = positional_encoding(conv(directions), spec.num_encoding_fn_dir, spec.include_input_dir,
                               spec.log_sampling_dir)
    feat = torch.relu(lin(h, w["fc_feat.weight"], w["fc_feat.bias"]))
    sigma = lin(h, w["fc_alpha.weight"], w["fc_alpha.bias"])
    v = torch.cat((feat, view), dim=-1)                          # feat first (models.py:72)
    v = torch.relu(lin(v, w["layers_dir.0.weight"], w["layers_dir.0.bias"]))
    rgb = torch.sigmoid(lin(v, w["fc_rgb.weight"], w["fc_rgb.bias"]))
    return torch.cat((rgb, sigma), dim=-1)


def coarse_intervals(near, far, count, ray_count, lindisp=False):
    """R1 (deterministic branch).  near/far: python floats, 0-dim or (R,) tensors."""
    u = torch.linspace(0.0, 1.0, count)[None, :]
    near = torch.as_tensor(near, dtype=torch.float32)
    far = torch.as_tensor(far, dtype=torch.float32)
    per_ray = near.dim() > 0 and near.shape[0] == ray_count
    if per_ray:
        near, far = near[:, None], far[:, None]
    if not lindisp:
        t = near * (1.0 - u) + far * u
    else:
        t = 1.0 / (1.0 / near * (1.0 - u) + 1.0 / far * u)
    if not per_ray:
        t = t.expand([ray_count, count])
    return t


def ray_points(t, directions, origins):
    """R2: p = o + d * t, (R,S,3)."""
    return origins[..., None, :] + directions[..., None, :] * t[..., :, None]


def exclusive_cumprod(x):
    """nerf_helpers.py:199-223"""
    c = torch.roll(torch.cumprod(x, -1), 1, -1)
    c[..., 0] = 1.0
    return c


def composite(radiance, t, directions, rs: RenderSpec, noise=None):
    """R4.  radiance (R,S,4), t (R,S), directions (R,3) -> dict of maps.  `noise` (R,S): the training-mode
    radiance noise, already scaled by its std (modules.py:82-91); differentiable in `radiance`."""
    big = torch.tensor([1e10])
    dists = torch.cat((t[..., 1:] - t[..., :-1], big.expand(t[..., :1].shape)), dim=-1)
    dists = dists * directions[..., None, :].norm(p=2, dim=-1)
    rgb = radiance[..., :3]
    sigma = torch.relu(radiance[..., 3] + (0.0 if noise is None else noise))
    alpha = 1.0 - torch.exp(-sigma * dists)
    trans = exclusive_cumprod(1.0 - alpha + 1e-10)
    mask = (trans > rs.attenuation_threshold).float()
    weights = alpha * trans
    rgb_map = (weights[..., None] * rgb).sum(dim=-2)
    acc = weights.sum(dim=-1)
    depth = (weights * t).sum(dim=-1)
    disp = 1.0 / torch.max(1e-10 * torch.ones_like(depth), depth / acc)
    disp[torch.isnan(disp)] = 0
    if not rs.training:
        depth = torch.where(acc < 1.0, torch.zeros_like(depth), depth)   # modules.py:108-109
    if rs.white_background:
        rgb_map = rgb_map + (1.0 - acc[..., None])
    return dict(rgb_map=rgb_map, depth_map=depth, weights=weights, mask_weights=mask,
                acc_map=acc, disp_map=disp)


def perturb_intervals(t, t_rand):
    """R1 with cfg.perturb (modules.py:171-184): stratified samples, t_rand (R,S) = the torch.rand draw."""
    mids = 0.5 * (t[..., 1:] + t[..., :-1])
    upper = torch.cat((mids, t[..., -1:]), dim=-1)
    lower = torch.cat((t[..., :1], mids), dim=-1)
    return lower + (upper - lower) * t_rand


def sample_pdf_intervals(t, weights, num_fine, u=None):
    """R5.  t (R,Sc), weights (R,Sc) -> sorted (R, Sc+num_fine).  u=None: deterministic linspace;
    u (R,num_fine): the torch.rand draw of the perturb branch (modules.py:224-228)."""
    bins = 0.5 * (t[..., 1:] + t[..., :-1])
    w = weights[..., 1:-1] + 1e-5
    pdf = w / torch.sum(w, dim=-1, keepdim=True)
    cdf = torch.cumsum(pdf, dim=-1)
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], dim=-1)
    if u is None:
        u = torch.linspace(0.0, 1.0, steps=num_fine).expand(list(cdf.shape[:-1]) + [num_fine])
    u = u.contiguous()
    cdf = cdf.contiguous()
    idx = torch.searchsorted(cdf, u, right=True)
    lo = torch.max(torch.zeros_like(idx), idx - 1)
    hi = torch.min((cdf.shape[-1] - 1) * torch.ones_like(idx), idx)
    both = torch.stack((lo, hi), dim=-1)
    shape = (both.shape[0], both.shape[1], cdf.shape[-1])
    cdf_g = torch.gather(cdf.unsqueeze(1).expand(shape), 2, both)
    bins_g = torch.gather(bins.unsqueeze(1).expand(shape), 2, both)
    denom = cdf_g[..., 1] - cdf_g[..., 0]
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    frac = (u - cdf_g[..., 0]) / denom
    new = bins_g[..., 0] + frac * (bins_g[..., 1] - bins_g[..., 0])
    merged, _ = torch.sort(torch.cat((t, new), dim=-1), dim=-1)
    return merged


def render(w_coarse, w_fine, spec_c: MLPSpec, spec_f, rs: RenderSpec, origins, directions, near, far):
    """R6: NeRFModel.forward.  Returns (coarse dict, fine dict | None); each dict also carries
    `t` (the sample depths used) and `radiance`, which the reference does not expose but the
    per-stage parity tests need."""
    origins, directions = _t(origins), _t(directions)
    R = directions.shape[0]
    t = coarse_intervals(near, far, rs.num_coarse, R, rs.lindisp)
    pts = ray_points(t, directions, origins)
    dirs = directions[..., None, :].expand_as(pts)
    rad = mlp_forward(w_coarse, spec_c, pts, dirs)
    coarse = composite(rad, t, directions, rs)
    coarse["t"], coarse["radiance"] = t, rad
    fine = None
    if w_fine is not None:
        tf = sample_pdf_intervals(t, coarse["weights"], rs.num_fine)
        pts = ray_points(tf, directions, origins)
        dirs = directions[..., None, :].expand_as(pts)
        radf = mlp_forward(w_fine, spec_f, pts, dirs)
        fine = composite(radf, tf, directions, rs)
        fine["t"], fine["radiance"] = tf, radf
    return coarse, fine


def query(*args, **kw):
    """NeRFModel.query (model_nerf.py:80-86): finest bundle."""
    c, f = render(*args, **kw)
    return f if f is not None else c


# ----------------------------------------------------------------------------- rays (R0)

def pose_spherical(theta, phi, radius):
    """data_helpers.py:8-37 (translate-z, rotate-phi about x, rotate-theta about y, axis flip)."""
    tr = np.eye(4, dtype=np.float32)
    tr[2, 3] = radius
    p = phi / 180.0 * np.pi
    rp = np.array([[1, 0, 0, 0], [0, np.cos(p), -np.sin(p), 0], [0, np.sin(p), np.cos(p), 0], [0, 0, 0, 1]],
                  dtype=np.float32)
    th = theta / 180.0 * np.pi
    rt = np.array([[np.cos(th), 0, -np.sin(th), 0], [0, 1, 0, 0], [np.sin(th), 0, np.cos(th), 0], [0, 0, 0, 1]],
                  dtype=np.float32)
    c2w = rt @ (rp @ tr)
    flip = np.array([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]], dtype=np.float32)
    return (flip @ c2w).astype(np.float32)


def get_ray_bundle(height, width, focal, c2w):
    """R0: returns origin (3,), directions (H,W,3) unit length, row-major (row=j, col=i)."""
    c2w = _t(c2w)
    ii, jj = torch.meshgrid(torch.arange(width, dtype=torch.float32),
                            torch.arange(height, dtype=torch.float32), indexing="ij")
    ii, jj = ii.transpose(-1, -2), jj.transpose(-1, -2)
    d = torch.stack([(ii - width * 0.5) / focal, -(jj - height * 0.5) / focal, -torch.ones_like(ii)], dim=-1)
    d = d / d.norm(2, dim=-1)[..., None]
    d = torch.sum(d[..., None, :] * c2w[:3, :3], dim=-1)
    return c2w[:3, -1], d


def ndc_rays(H, W, focal, near, rays_o, rays_d):
    """nerf_helpers.py:280-307"""
    t = -(near + rays_o[..., 2]) / rays_d[..., 2]
    rays_o = rays_o + t[..., None] * rays_d
    o0 = -1.0 / (W / (2.0 * focal)) * rays_o[..., 0] / rays_o[..., 2]
    o1 = -1.0 / (H / (2.0 * focal)) * rays_o[..., 1] / rays_o[..., 2]
    o2 = 1.0 + 2.0 * near / rays_o[..., 2]
    d0 = -1.0 / (W / (2.0 * focal)) * (rays_d[..., 0] / rays_d[..., 2] - rays_o[..., 0] / rays_o[..., 2])
    d1 = -1.0 / (H / (2.0 * focal)) * (rays_d[..., 1] / rays_d[..., 2] - rays_o[..., 1] / rays_o[..., 2])
    d2 = -2.0 * near / rays_o[..., 2]
    return torch.stack([o0, o1, o2], -1), torch.stack([d0, d1, d2], -1)


# ----------------------------------------------------------------------------- eval bookkeeping (R8)

def view_loss(rgb, target, chunk):
    """eval_nerf.py:56-76: fp32 tensor sum of per-chunk mse / FLOAT batch_count
    (640000/2048 = 312.5 although 313 chunks run)."""
    n = rgb.shape[0]
    batch_count = n / chunk
    loss = 0
    for s in range(0, n, chunk):
        loss += torch.nn.functional.mse_loss(rgb[s:s + chunk], target[s:s + chunk])
    loss /= batch_count
    return loss


def mse2psnr(mse):
    """nerf_helpers.py:17-23 (0-dim tensor in, 0-dim tensor out)."""
    mse = torch.as_tensor(mse)
    if mse == 0:
        mse = torch.tensor(1e-5)
    return -10.0 * torch.log10(mse)


def dataset_loss(view_losses):
    """eval_nerf.py:104: mean over views of the per-view losses."""
    return torch.stack([torch.as_tensor(v) for v in view_losses]).mean()


# ----------------------------------------------------------------------------- dense grid (R11)

def grid_points(limit, res):
    """mesh_nerf.py:37-40: linspace per axis, meshgrid ij, last axis fastest."""
    nums = (res,) * 3 if isinstance(res, int) else tuple(res)
    tiles = [torch.linspace(-limit, limit, n) for n in nums]
    return torch.stack(torch.meshgrid(*tiles, indexing="ij"), -1).view(-1, 3).float()


def extract_radiance(w, spec, limit, res, batch=65536):
    """mesh_nerf.py:27-53: sample_points(p, p) -- the points themselves are the view dirs."""
    pts = grid_points(limit, res)
    out = [mlp_forward(w, spec, pts[s:s + batch], pts[s:s + batch]) for s in range(0, pts.shape[0], batch)]
    nums = (res,) * 3 if isinstance(res, int) else tuple(res)
    return torch.cat(out, 0).view(*nums, 4).contiguous().numpy()


def iso_level(density, requested):
    """mesh_nerf.py:56-65 on a numpy fp32 volume."""
    lo, hi, sd = density.min(), density.max(), density.std()
    return min(max(requested, lo + sd), hi - sd)


# ----------------------------------------------------------------------------- BuFF voxel tree (R9, R10)

def buff_initial_voxels(near, far, outer_count):
    """TreeSampling.__init__ + Node.subdivide + consolidate for a fresh tree (tree.py:71-92,19-33,162-175):
    outer_count^3 voxels tiling [near - mean, far - mean]^3, x-major / z-fastest, (N,2,3) min/max corners."""
    mean = (near + far) / 2
    lo, hi = torch.tensor([near - mean] * 3), torch.tensor([far - mean] * 3)
    extent = hi - lo
    out = []
    for i in range(outer_count):
        for g in range(outer_count):
            for h in range(outer_count):
                a = torch.tensor([i, g, h], dtype=torch.float) / outer_count * extent
                b = torch.tensor([i + 1, g + 1, h + 1], dtype=torch.float) / outer_count * extent
                out.append(torch.stack((lo + a, lo + b), 0))
    return torch.stack(out, 0)


def buff_intersect(voxels, origins, dirs, near, far, samples_count, ties="stable"):
    """TreeSampling.batch_ray_voxel_intersect, deterministic branch (tree.py:215-343).
    Returns (z_vals (R,S), voxel indices (R,S) int64, ray_mask (R,) bool).
    ties="stable": the three sorts are stable, so every id is the voxel its sample lies in.
    ties="reference": the sorts are issued exactly as the reference issues them (torch.sort's unstable default --
    on the CPU libstdc++'s introsort over (key, index) pairs, a deterministic algorithm, see oracle/introsort.py):
    the ids equal the reference's bit for bit on the same torch build (tests/golden/buff_fern.npz)."""
    stable = {"stable": True, "reference": False}[ties]
    voxels, origins, dirs = _t(voxels), _t(origins), _t(dirs)
    R, N = dirs.shape[0], voxels.shape[0]
    inv = 1 / dirs
    signs = (inv < 0).long()
    b = voxels.transpose(0, 1)                                     # (2,N,3)
    ax = torch.arange(3)

    def pick(s):                                                   # (R,N,3): bounds[s[r,a], n, a]
        return b[s[:, None, :].expand(R, N, 3), torch.arange(N)[None, :, None].expand(R, N, 3), ax[None, None, :].expand(R, N, 3)]

    o = origins[:, None, :]
    tvmin = (pick(signs) - o) * inv[:, None, :]
    tvmax = (pick(1 - signs) - o) * inv[:, None, :]
    mask = (tvmin[..., 0] <= tvmax[..., 1]) & (tvmin[..., 1] <= tvmax[..., 0])
    tmin = torch.where(tvmin[..., 1] > tvmin[..., 0], tvmin[..., 1], tvmin[..., 0])
    tmax = torch.where(tvmax[..., 1] < tvmax[..., 0], tvmax[..., 1], tvmax[..., 0])
    mask = mask & (tmin <= tvmax[..., 2]) & (tvmin[..., 2] <= tmax)
    tmin = torch.where(tvmin[..., 2] > tmin, tvmin[..., 2], tmin)
    tmax = torch.where(tvmax[..., 2] < tmax, tvmax[..., 2], tmax)
    mask = mask & (tmin >= near) & (tmax <= far)
    ray_mask = mask.sum(-1) > 0
    order = torch.sort(tmin, dim=-1, stable=stable)
    inter = torch.stack((tmin, tmax), -1).gather(-2, order.indices[..., None].expand(R, N, 2))
    mask_sorted = mask.gather(-1, order.indices)
    start = torch.sort(mask_sorted.long(), dim=-1, descending=True, stable=stable)
    res = torch.zeros_like(inter)
    res[start.values.bool()] = inter[mask_sorted]
    cums = torch.cumsum(res[..., 1] - res[..., 0], -1)
    samples = torch.linspace(0, 1.0, samples_count) * cums[..., -1][..., None]
    bucket = torch.searchsorted(cums, samples)
    first = torch.searchsorted(bucket, bucket, right=False)
    offset = samples - samples.gather(-1, first)
    z = res[..., 0].gather(-1, bucket) + offset
    vox = order.indices.gather(-1, start.indices.gather(-1, bucket))
    z, zorder = torch.sort(z, dim=-1, stable=stable)
    return z, vox.gather(-1, zorder), ray_mask


def _buff_slab_test(voxels, origins, dirs, near, far):
    """The slab test both sampler branches share (tree.py:226-271): (tmin, tmax, mask), each (R,N)."""
    voxels, origins, dirs = _t(voxels), _t(origins), _t(dirs)
    R, N = dirs.shape[0], voxels.shape[0]
    inv = 1 / dirs
    signs = (inv < 0).long()
    b = voxels.transpose(0, 1)
    ax = torch.arange(3)

    def pick(s):
        return b[s[:, None, :].expand(R, N, 3), torch.arange(N)[None, :, None].expand(R, N, 3), ax[None, None, :].expand(R, N, 3)]

    o = origins[:, None, :]
    tvmin = (pick(signs) - o) * inv[:, None, :]
    tvmax = (pick(1 - signs) - o) * inv[:, None, :]
    mask = (tvmin[..., 0] <= tvmax[..., 1]) & (tvmin[..., 1] <= tvmax[..., 0])
    tmin = torch.where(tvmin[..., 1] > tvmin[..., 0], tvmin[..., 1], tvmin[..., 0])
    tmax = torch.where(tvmax[..., 1] < tvmax[..., 0], tvmax[..., 1], tvmax[..., 0])
    mask = mask & (tmin <= tvmax[..., 2]) & (tvmin[..., 2] <= tmax)
    tmin = torch.where(tvmin[..., 2] > tmin, tvmin[..., 2], tmin)
    tmax = torch.where(tvmax[..., 2] < tmax, tvmax[..., 2], tmax)
    mask = mask & (tmin >= near) & (tmax <= far)
    return tmin, tmax, mask


def buff_intersect_random(voxels, origins, dirs, near, far, u_pick, u_pos):
    """TreeSampling.batch_ray_voxel_intersect, `tree.use_random_sampling` branch (tree.py:280-297, :337-341), as a
    function of the draws: u_pick (R,S) float64 = what `torch.multinomial(weights, S, replacement=True)` draws,
    u_pos (R,S) float32 = `torch.rand_like(values_min)`.
    torch.multinomial with replacement on the CPU (ATen/native/cpu/MultinomialKernel.cpp) is an inverse-CDF sampler:
    per row the fp32 running sum of the weights divided by its total (last entry forced to 1), then per sample ONE
    double from the generator and a lower-bound search -- the first category whose cumulative probability is >= u.
    With weights 1 (crossed voxel) / 1e-12 (others) the running sum is the count of crossed voxels so far (1e-12 is
    absorbed once the sum is >= 1), so the pick is the ceil(u K)-th crossed voxel in index order, decided in fp32
    (`count / K` rounded).  Under one seed the reference's draws ARE `torch.rand(R * S, dtype=float64)` followed by
    `torch.rand(R, S)`: tests/golden/buff_random.npz pins this function to the unmodified reference bit for bit.
    Returns (z (R,S), voxel ids (R,S) int64, ray_mask (R,)); rows of rays that cross no voxel are unspecified (the
    reference samples all voxels there and BuFFModel.forward overwrites those rows, model_buff.py:51-53)."""
    tmin, tmax, mask = _buff_slab_test(voxels, origins, dirs, near, far)
    ray_mask = mask.sum(-1) > 0
    weights = torch.ones_like(tmin)
    weights[~mask] = 1e-12
    cum = torch.cumsum(weights, -1)                    # fp64 accumulation rounded per element: the counts, exactly
    cum = cum / cum[..., -1:]
    cum[..., -1] = 1.0
    pick = torch.searchsorted(cum.double(), torch.as_tensor(np.asarray(u_pick), dtype=torch.float64).contiguous(), right=False).clamp_(max=tmin.shape[1] - 1)
    vmin, vmax = tmin.gather(-1, pick), tmax.gather(-1, pick)
    z = vmin + (vmax - vmin) * _t(u_pos)
    z, order = torch.sort(z, dim=-1, stable=True)
    return z, pick.gather(-1, order), ray_mask


def buff_integrate(memm, counter, indices, weights, mask_weights):
    """(f)-3: TreeSampling.ray_batch_integration (tree.py:177-206): returns the updated voxel weights.
    indices (K,S) int64, weights / mask_weights (K,S): the rays that hit the tree."""
    rays, n = weights.shape[0], memm.shape[0]
    acc = torch.zeros(rays, n).scatter_add(-1, indices, weights).sum(0)
    freq = torch.zeros(rays, n).scatter_add(-1, indices, mask_weights).sum(0)
    seen = freq > 0
    out = memm.clone()
    out[seen] += (acc[seen] / freq[seen] - out[seen]) / counter
    return out
