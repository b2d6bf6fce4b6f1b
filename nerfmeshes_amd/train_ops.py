"""Training-mode host wrappers over the training entry points of the C ABI.

What `loss.backward()` does in the reference's `NeRFModel.training_step`
(/root/reference/src/models/model_nerf.py:88-151) is split here as the library is:

* `mlp_rays`       -- FlexibleNeRFModel.forward over ray samples as a `torch.autograd.Function`:
                      forward = the fused HIP kernel recording a tape, backward = the HIP delta kernel +
                      the hand-written weight-gradient kernels (dW = delta^T @ activation rows: nm_weight_grad_ex,
                      nm_head_grad_ex for fc_alpha / fc_rgb / fc_out) -- the kernels tuned for the shipped configs'
                      128- / 256-wide layers when they apply, the general ones (nerf_dw_g.hip: any width, any stride,
                      any sample count) otherwise.  No library GEMM and no library reduction is left on any shape
                      nm_mlp_create accepts: tuned kernels for the shipped configs' shapes, the generic family otherwise.
* `composite`      -- VolumeRenderer.forward (noise + ReLU + alpha compositing) with a HIP backward.
* `perturb_intervals`, `sample_pdf_rand` -- the stochastic depth samplers; random numbers are torch's.

torch is plumbing (device memory, autograd bookkeeping, the optimizer); there is no CPU path.
"""
import ctypes as C
import os

import torch

from . import _lib, hip_ops
from ._lib import BundleGrads, MlpDeltas, MlpParamGrads, MlpTape, MlpWeights, WeightGradJob, check
from .hip_ops import _dev32, _ptr, _stream


_STAGES = None     # bench.py's per-stage breakdown: when a dict, name -> [(start event, end event), ...] on torch's current stream


class _stage:
    """`with _stage("delta"): ...` -- HIP events around a stage of the training step while `profile_stages(True)` is on."""

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        if _STAGES is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record()

    def __exit__(self, *exc):
        if _STAGES is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            _STAGES.setdefault(self.name, []).append((self.e0, e1))
        return False


def profile_stages(on):
    """Start (True) / stop (False) collecting per-stage HIP-event times; stopping returns {stage: total ms} since the start."""
    global _STAGES
    if on:
        _STAGES = {}
        return None
    torch.cuda.synchronize()
    done, _STAGES = _STAGES or {}, None
    return {k: sum(a.elapsed_time(b) for a, b in v) for k, v in done.items()}


def param_names(num_layers, use_viewdirs=True):
    names = ["layer1.weight", "layer1.bias"]
    for i in range(num_layers - 1):
        names += [f"layers_xyz.{i}.weight", f"layers_xyz.{i}.bias"]
    if not use_viewdirs:                       # models.py:52-55: the trunk ends in fc_out (4, H)
        return names + ["fc_out.weight", "fc_out.bias"]
    return names + ["fc_feat.weight", "fc_feat.bias", "fc_alpha.weight", "fc_alpha.bias", "layers_dir.0.weight",
                    "layers_dir.0.bias", "fc_rgb.weight", "fc_rgb.bias"]


def _weights_struct(mlp, params):
    """nm_mlp_weights over live device tensors (dict name -> CUDA fp32 tensor) + the tensors it points into (keep alive)."""
    L = int(mlp.desc["num_layers"])
    if not mlp.desc.get("use_viewdirs", True):
        # models.py:77-79: fc_out (4, H) supplies the colour rows (0..2) and the density row (3), see nm_mlp_weights
        names = ["layer1.weight", "layer1.bias", "fc_out.weight", "fc_out.bias"]
        names += [f"layers_xyz.{i}.{k}" for i in range(L - 1) for k in ("weight", "bias")]
        keep = {k: _dev32(params[k], mlp.device, k) for k in names}
        p = lambda k: C.c_void_p(keep[k].data_ptr())  # noqa: E731
        H = int(mlp.desc["hidden_size"])
        xs_w = (C.c_void_p * (L - 1))(*[p(f"layers_xyz.{i}.weight") for i in range(L - 1)])
        xs_b = (C.c_void_p * (L - 1))(*[p(f"layers_xyz.{i}.bias") for i in range(L - 1)])
        ow, ob = keep["fc_out.weight"].data_ptr(), keep["fc_out.bias"].data_ptr()
        w = MlpWeights(p("layer1.weight"), p("layer1.bias"), xs_w, xs_b, None, None, C.c_void_p(ow + 3 * H * 4),
                       C.c_void_p(ob + 3 * 4), C.c_void_p(ow), C.c_void_p(ob), None, None, None, None)
        return w, (keep, xs_w, xs_b)
    keep = {k: _dev32(params[k], mlp.device, k) for k in param_names(L)}
    p = lambda k: C.c_void_p(keep[k].data_ptr())  # noqa: E731
    xs_w = (C.c_void_p * (L - 1))(*[p(f"layers_xyz.{i}.weight") for i in range(L - 1)])
    xs_b = (C.c_void_p * (L - 1))(*[p(f"layers_xyz.{i}.bias") for i in range(L - 1)])
    w = MlpWeights(p("layer1.weight"), p("layer1.bias"), xs_w, xs_b, p("layers_dir.0.weight"), p("layers_dir.0.bias"),
                   p("fc_alpha.weight"), p("fc_alpha.bias"), p("fc_rgb.weight"), p("fc_rgb.bias"),
                   p("fc_feat.weight"), p("fc_feat.bias"), None, None)
    return w, (keep, xs_w, xs_b)


def refresh(mlp, params, struct=None):
    """Re-pack `mlp` (hip_ops.HipMLP) from live device tensors: dict name -> CUDA fp32 tensor, or a (MlpWeights, keep-alive)
    pair `_weights_struct` built earlier over the same storages.  Returns that pair (for the caller to cache)."""
    w, keep = struct if struct is not None else _weights_struct(mlp, params)
    check(_lib.load().nm_mlp_refresh(mlp.handle, C.byref(w), _stream()), "nm_mlp_refresh")
    return w, keep


def _per_ray(origins, rays):
    return int(origins.reshape(-1, 3).shape[0] == rays and rays > 1)


def forward_train(mlp, origins, dirs, t):
    """-> (radiance (R,S,4), tape dict of device tensors)."""
    lib = _lib.load()
    origins, dirs, t = (_dev32(x, mlp.device) for x in (origins, dirs, t))
    rays, samples = t.shape
    n, H, L = rays * samples, int(mlp.desc["hidden_size"]), int(mlp.desc["num_layers"])
    tiles = (n + 15) // 16
    f32 = dict(dtype=torch.float32, device=mlp.device)
    flat = not mlp.desc.get("use_viewdirs", True)      # trunk-only tape: no fc_feat / layers_dir activations
    generic = mlp.kernel_variant()[0] >= 1000          # generic-shape family: ReLU' is read off the activation rows, no masks
    tape_bytes = 4 * n * (L * H + (0 if flat else H + H // 2))
    if 2.1 * tape_bytes > torch.cuda.get_device_properties(mlp.device).total_memory:     # tape + deltas of the backward
        raise _lib.HipLibraryError(
            f"the training tape of {rays} rays x {samples} samples needs {tape_bytes / 2**30:.0f} GiB (+ as much for the "
            "deltas): use a smaller ray chunk, or torch.no_grad() if this is inference")
    tape = dict(h=torch.empty(L, n, H, **f32), feat=None if flat else torch.empty(n, H, **f32), v=None,
                mask_h=None if generic else torch.empty(L, tiles, 64, dtype=torch.int64, device=mlp.device),
                mask_v=None if flat or generic else torch.empty(tiles, 64, dtype=torch.int64, device=mlp.device),
                enc_x=None, enc_d=None)
    if lib.nm_mlp_tapes_encodings(mlp.handle):
        # tuned family: the kernel writes the encoding rows the layer1 / skip / view weight gradients contract with (it has
        # them in registers); the backward then skips its nm_encode_samples_strided pass
        tape["enc_x"] = torch.empty(n, 64, **f32)
        tape["enc_d"] = None if flat else torch.empty(n, 64, **f32)
        if lib.nm_mlp_backward_fused_supported(mlp.handle, n):
            # 64-wide networks: the view layer's 32-float activation rows live in the unused half of the direction-encoding rows
            # (27 floats of 64), so that one block of rows feeds layers_dir[0]'s and fc_rgb's gradients in nm_mlp_backward_fused
            tape["v"] = tape["enc_d"][:, 32:]
    if tape["v"] is None and not flat:
        tape["v"] = torch.empty(n, H // 2, **f32)
    # layer1's output (h[0]) is read by no backward that takes layer1's and layers_xyz[0]'s gradients by linearity: not written then
    # (a tenth of the tape's bytes).  Only where that backward is available whatever the sample count: backward() falls back to it
    tape["h0_taped"] = not (bool(lib.nm_mlp_backward_stops_at_xyz0(mlp.handle)) and
                            (n * H * H > LINEAR_LAYER1_MIN_WORK or bool(lib.nm_mlp_backward_fused_supported(mlp.handle, n))))
    out = torch.empty(rays, samples, 4, **f32)
    ct = _tape_struct(tape)
    with _stage("taping_forward"):
        check(lib.nm_mlp_forward_train(mlp.handle, _ptr(origins), _per_ray(origins, rays), _ptr(dirs), _ptr(t), rays, samples,
                                       C.byref(ct), _ptr(out), _stream()), "nm_mlp_forward_train")
    return out, tape


def _tape_struct(tape):
    opt = lambda x: None if x is None else _ptr(x)   # noqa: E731
    v = tape["v"]
    return MlpTape(_ptr(tape["h"]), opt(tape["feat"]), opt(v), opt(tape["mask_h"]), opt(tape["mask_v"]),
                   opt(tape.get("enc_x")), opt(tape.get("enc_d")), 0 if v is None else int(v.stride(0)),
                   0 if tape.get("h0_taped", True) else 1)


def encode_samples(mlp, origins, dirs, t):
    """PositionalEncoding rows of the sample points / view directions: ((n, dim_xyz), (n, dim_dir))."""
    origins, dirs, t = (_dev32(x, mlp.device) for x in (origins, dirs, t))
    rays, samples = t.shape
    d = mlp.desc
    dx = 6 * int(d["num_encoding_fn_xyz"]) + (3 if d.get("include_input_xyz", True) else 0)
    dd = 6 * int(d["num_encoding_fn_dir"]) + (3 if d.get("include_input_dir", True) else 0)
    ex = torch.empty(rays * samples, dx, dtype=torch.float32, device=mlp.device)
    ed = torch.empty(rays * samples, dd, dtype=torch.float32, device=mlp.device)
    check(_lib.load().nm_encode_samples(mlp.handle, _ptr(origins), _per_ray(origins, rays), _ptr(dirs), _ptr(t), rays,
                                        samples, _ptr(ex), _ptr(ed), _stream()), "nm_encode_samples")
    return ex, ed


_dw_ws = {}
LINEAR_LAYER1_MIN_WORK = 4e9      # samples x hidden^2 from which backward() takes layer1's gradient by linearity (tests set it to 0)


def _workspace(mlp, tag, need):
    key = (mlp.device, torch.cuda.current_stream(mlp.device).cuda_stream, tag)
    ws = _dw_ws.get(key)
    if ws is None or ws.numel() < need:
        ws = _dw_ws[key] = torch.empty(need, dtype=torch.uint8, device=mlp.device)
    return ws


def _weight_grad(mlp, delta, act, in_features, out=None, col0=0, bias=True):
    """dW = delta^T @ act[:, :in_features] (+ column sums of delta) through nm_weight_grad_ex: fp32 MFMA, split over the
    samples across the CUs, order-fixed reduction.  delta (n, out) / act (n, stride >= in_features): row-major views of
    ANY width, stride and row count -- the kernel tuned for the shipped configs' shapes when it applies, the general
    one (nerf_dw_g.hip) otherwise; never a library GEMM."""
    lib = _lib.load()
    n, o = delta.shape
    assert delta.stride(1) == 1 and act.stride(1) == 1 and act.shape[0] == n
    lda, ldb = delta.stride(0), act.stride(0)
    cus = int(lib.nm_mlp_num_cus(mlp.handle))
    ws = _workspace(mlp, "dw", int(lib.nm_weight_grad_workspace_bytes_ex(o, lda, in_features, ldb, cus)))
    if out is None:
        out = torch.empty(o, in_features, dtype=torch.float32, device=mlp.device)
    db = torch.empty(o, dtype=torch.float32, device=mlp.device) if bias else None
    with _stage("weight_gradients"):
        check(lib.nm_weight_grad_ex(cus, _ptr(delta), o, lda, _ptr(act), in_features, ldb, n, _ptr(ws), _ptr(out), out.shape[1],
                                    col0, _ptr(db), _stream()), "nm_weight_grad_ex")
    return out, db


def _weight_grad_batch(mlp, jobs):
    """Several dW products in as few launches as possible: `jobs` = [(delta, act, in_features, out, col0, bias tensor | None)];
    products of one shape, stride pair and row count go through nm_weight_grad_batch together (one launch + one reduction,
    1 / len of the partials each: the L same-shape layers of a network), in groups of at most 16."""
    lib = _lib.load()
    cus = int(lib.nm_mlp_num_cus(mlp.handle))
    groups = {}
    for job in jobs:
        delta, act, in_features = job[0], job[1], job[2]
        assert delta.stride(1) == 1 and act.stride(1) == 1 and act.shape[0] == delta.shape[0]
        groups.setdefault((delta.shape[1], delta.stride(0), in_features, act.stride(0), delta.shape[0]), []).append(job)
    for (o, lda, in_features, ldb, n), group in groups.items():
        ws = _workspace(mlp, "dw", int(lib.nm_weight_grad_workspace_bytes_ex(o, lda, in_features, ldb, cus)))
        for s0 in range(0, len(group), 16):
            part = group[s0:s0 + 16]
            arr = (WeightGradJob * len(part))(*[WeightGradJob(_ptr(d), _ptr(a), _ptr(out), out.shape[1], col0, _ptr(db))
                                                 for d, a, _, out, col0, db in part])
            with _stage("weight_gradients"):
                check(lib.nm_weight_grad_batch(cus, len(part), arr, o, lda, in_features, ldb, n, _ptr(ws), _stream()), "nm_weight_grad_batch")


def _head_grad(mlp, dlast, act, bias=False):
    """(4, K) = dlast^T @ act for the heads that share dlast (n, 4) (+ its column sums): nm_head_grad_ex, HBM-bound VALU
    kernel with the order-fixed partial reduction; any activation width."""
    lib = _lib.load()
    n, k = act.shape
    ws = _workspace(mlp, "head", int(lib.nm_head_grad_workspace_bytes_ex(k)))
    out = torch.empty(4, k, dtype=torch.float32, device=mlp.device)
    db = torch.empty(4, dtype=torch.float32, device=mlp.device) if bias else None
    with _stage("head_gradients"):
        check(lib.nm_head_grad_ex(_ptr(dlast), _ptr(act), k, act.stride(0), n, _ptr(ws), _ptr(out), _ptr(db), _stream()),
              "nm_head_grad_ex")
    return out, db


def backward(mlp, tape, radiance, grad_radiance, origins, dirs, t):
    """Parameter gradients of sum(radiance * grad_radiance): dict keyed like FlexibleNeRFModel.state_dict()."""
    lib = _lib.load()
    radiance, grad_radiance = _dev32(radiance, mlp.device), _dev32(grad_radiance, mlp.device)
    d = mlp.desc
    L, H, skip_step = int(d["num_layers"]), int(d["hidden_size"]), int(d["skip_step"])
    n = radiance.numel() // 4
    f32 = dict(dtype=torch.float32, device=mlp.device)
    flat = not d.get("use_viewdirs", True)
    if (tape.get("enc_d") is not None and tape["v"] is not None and tape["v"].data_ptr() == tape["enc_d"].data_ptr() + 128
            and lib.nm_mlp_backward_fused_supported(mlp.handle, n)):
        return _backward_fused(mlp, tape, radiance, grad_radiance, n)
    dh, dlast = torch.empty(L, n, H, **f32), torch.empty(n, 4, **f32)
    dfeat, dv = (None, None) if flat else (torch.empty(n, H, **f32), torch.empty(n, H // 2, **f32))
    ct = _tape_struct(tape)
    cd = MlpDeltas(_ptr(dh), None if flat else _ptr(dfeat), None if flat else _ptr(dv), _ptr(dlast))
    # layer1 has no activation (models.py:62): the delta at its output is W0^T times the delta at layers_xyz[0], so its gradient is
    # W0^T applied ONCE to sums over the samples -- the tuned delta kernels then stop one transposed layer early (an eighth of the
    # kernel at 8 layers) and d_h[0] is never produced
    # (worth it from ~60 us of saved matrix work on: the path costs five small launches -- n H^2 > 4e9)
    linear_l1 = bool(lib.nm_mlp_backward_stops_at_xyz0(mlp.handle)) and (n * H * H > LINEAR_LAYER1_MIN_WORK or not tape.get("h0_taped", True))
    if not linear_l1 and not tape.get("h0_taped", True):
        raise _lib.HipLibraryError("this tape was taken without layer1's output (h[0]) for a backward that takes its gradients by linearity, "
                                   "which has been switched off since (NM_BACKWARD_LINEAR_LAYER1=0 between forward and backward?)")
    with _stage("delta"):
        check(lib.nm_mlp_backward_ex(mlp.handle, n, C.byref(ct), _ptr(radiance), _ptr(grad_radiance), C.byref(cd), 1 if linear_l1 else 0,
                                     _stream()), "nm_mlp_backward_ex")
    h, feat, v = tape["h"], tape["feat"], tape["v"]
    dx = 6 * int(d["num_encoding_fn_xyz"]) + (3 if d.get("include_input_xyz", True) else 0)
    dd = 0 if flat else 6 * int(d["num_encoding_fn_dir"]) + (3 if d.get("include_input_dir", True) else 0)
    is_skip = lambda i: i % skip_step == 0 and i > 0 and i != L - 1   # noqa: E731  (cat(x, xyz): models.py:64-65)
    # the encoding rows the layer1 / skip / view gradients contract with: 64-float rows (what the tuned weight-gradient
    # kernel streams; padding written) when both encodings fit, rows of the next multiple of 4 floats otherwise (16-byte
    # DMA pieces for the general kernel; whatever lies beyond the width only reaches dW entries that are never read)
    origins, dirs, t = (_dev32(x, mlp.device) for x in (origins, dirs, t))
    rays, samples = t.shape
    if tape.get("enc_x") is not None:          # written by the taping forward (tuned family)
        enc_x, enc_d = tape["enc_x"], tape.get("enc_d")
    else:
        sx, sd = (64, 64) if dx <= 64 and dd <= 64 else ((dx + 3) & ~3, (max(dd, 1) + 3) & ~3)
        enc_x, enc_d = torch.empty(n, sx, **f32), (torch.empty(n, sd, **f32) if dd else None)
        with _stage("encodings"):
            check(lib.nm_encode_samples_strided(mlp.handle, _ptr(origins), _per_ray(origins, rays), _ptr(dirs), _ptr(t), rays,
                                                samples, _ptr(enc_x), sx, _ptr(enc_d), sd, _stream()), "nm_encode_samples_strided")
    # every dW product of the backward as a job (delta, act, in_features, out, col0, bias): the same-shape ones -- the hidden x hidden
    # layers, the two encoding products -- each go out as ONE launch (nm_weight_grad_batch)
    g, jobs = {}, []

    def product(name, delta, act, in_features, out=None, col0=0, bias=True):
        o = delta.shape[1]
        if out is None:
            out = torch.empty(o, in_features, **f32)
        db = torch.empty(o, **f32) if bias else None
        jobs.append((delta, act, in_features, out, col0, db))
        if bias:
            g[name + ".bias"] = db
        g[name + ".weight"] = out

    if linear_l1:
        l1_sums = torch.empty(H, dx, **f32)               # d_h[1]^T @ encoding; the column sums of d_h[1] = layers_xyz[0]'s bias gradient
        g["layers_xyz.0.bias"] = torch.empty(H, **f32)     # the column sums, from this product (layers_xyz[0]'s own is not run)
        jobs.append((dh[1], enc_x, dx, l1_sums, 0, g["layers_xyz.0.bias"]))
    else:
        product("layer1", dh[0], enc_x, dx)
    for i in range(L - 1):
        if linear_l1 and i == 0:
            continue          # grad(layers_xyz[0].weight) = [d_h[1]^T enc | sum d_h[1]] @ [W1 | b1]^T below: h[0] = layer1(enc) is linear in enc
        delta = dh[1 + i]
        if is_skip(i):                                                                  # cat(x, xyz): models.py:64-65
            gw = torch.empty(H, H + dx, **f32)
            product(f"layers_xyz.{i}", delta, h[i], H, out=gw, col0=0)
            product(f"layers_xyz.{i}", delta, enc_x, dx, out=gw, col0=H, bias=False)
        else:
            product(f"layers_xyz.{i}", delta, h[i], H)
    if not flat:
        product("fc_feat", dfeat, h[L - 1], H)
        gw = torch.empty(H // 2, H + dd, **f32)                                        # cat(feat, view): models.py:72
        product("layers_dir.0", dv, feat, H, out=gw, col0=0)
        if dd:
            product("layers_dir.0", dv, enc_d, dd, out=gw, col0=H, bias=False)
    _weight_grad_batch(mlp, jobs)
    if linear_l1:
        # with S = [ d_h[1]^T enc | sum d_h[1] ]:  grad(layer1) = W0^T S  and  grad(layers_xyz[0].weight) = d_h[1]^T @ h[0] = S [W1 | b1]^T
        # (h[0] = W1 enc + b1): one small kernel that reads W0, W1, b1 out of the handle's packed image -- what the forward of this
        # step used -- instead of a transposed layer of the delta chain and a hidden x hidden product, both over all n samples
        g["layer1.weight"], g["layer1.bias"], g["layers_xyz.0.weight"] = torch.empty(H, dx, **f32), torch.empty(H, **f32), torch.empty(H, H, **f32)
        check(lib.nm_mlp_linear_layer1_finish(mlp.handle, _ptr(l1_sums), dx, _ptr(g["layers_xyz.0.bias"]), _ptr(g["layer1.weight"]),
                                              _ptr(g["layer1.bias"]), _ptr(g["layers_xyz.0.weight"]), _stream()), "nm_mlp_linear_layer1_finish")
    # the 1-row / 3-row heads share dlast (n,4): one product per operand, rows picked afterwards (an MFMA tile would
    # waste 12 of its 16 rows; the product is HBM-bound on reading h / v once)
    if flat:
        # fc_out (4, H) over the trunk output: rows 0..2 take the pre-sigmoid colour deltas, row 3 the density delta --
        # exactly dlast^T @ h[L-1] and dlast's column sums (models.py:77-79)
        g["fc_out.weight"], g["fc_out.bias"] = _head_grad(mlp, dlast, h[L - 1], bias=True)
        return g
    (gh, last_sums), (gv, _) = _head_grad(mlp, dlast, h[L - 1], bias=True), _head_grad(mlp, dlast, v)
    g["fc_alpha.weight"], g["fc_alpha.bias"] = gh[3:4], last_sums[3:4]
    g["fc_rgb.weight"], g["fc_rgb.bias"] = gv[:3], last_sums[:3]
    return g


def _backward_fused(mlp, tape, radiance, grad_radiance, n):
    """The 64-wide networks (BASELINE config 1's 4x64): delta chain and every trunk / view-layer weight gradient in ONE kernel
    (nm_mlp_backward_fused, nerf_bwd_fused.hip), the two 4-row heads included: no delta row is written, the tape is read once."""
    lib = _lib.load()
    d = mlp.desc
    L, H, skip_step = int(d["num_layers"]), int(d["hidden_size"]), int(d["skip_step"])
    f32 = dict(dtype=torch.float32, device=mlp.device)
    dx = 6 * int(d["num_encoding_fn_xyz"]) + (3 if d.get("include_input_xyz", True) else 0)
    dd = 6 * int(d["num_encoding_fn_dir"]) + (3 if d.get("include_input_dir", True) else 0)
    is_skip = lambda i: i % skip_step == 0 and i > 0 and i != L - 1   # noqa: E731  (cat(x, xyz): models.py:64-65)
    g = {"layer1.weight": torch.empty(H, dx, **f32), "layer1.bias": torch.empty(H, **f32),
         "fc_feat.weight": torch.empty(H, H, **f32), "fc_feat.bias": torch.empty(H, **f32),
         "layers_dir.0.weight": torch.empty(H // 2, H + dd, **f32), "layers_dir.0.bias": torch.empty(H // 2, **f32)}
    pg = MlpParamGrads()
    for i in range(L - 1):
        g[f"layers_xyz.{i}.weight"] = torch.empty(H, H + (dx if is_skip(i) else 0), **f32)
        g[f"layers_xyz.{i}.bias"] = torch.empty(H, **f32)
        pg.xyz_weight[i], pg.xyz_bias[i] = _ptr(g[f"layers_xyz.{i}.weight"]), _ptr(g[f"layers_xyz.{i}.bias"])
    pg.layer1_weight, pg.layer1_bias = _ptr(g["layer1.weight"]), _ptr(g["layer1.bias"])
    pg.feat_weight, pg.feat_bias = _ptr(g["fc_feat.weight"]), _ptr(g["fc_feat.bias"])
    pg.dir_weight, pg.dir_bias = _ptr(g["layers_dir.0.weight"]), _ptr(g["layers_dir.0.bias"])
    g.update({"fc_alpha.weight": torch.empty(1, H, **f32), "fc_alpha.bias": torch.empty(1, **f32),
              "fc_rgb.weight": torch.empty(3, H // 2, **f32), "fc_rgb.bias": torch.empty(3, **f32)})
    pg.alpha_weight, pg.alpha_bias = _ptr(g["fc_alpha.weight"]), _ptr(g["fc_alpha.bias"])
    pg.rgb_weight, pg.rgb_bias = _ptr(g["fc_rgb.weight"]), _ptr(g["fc_rgb.bias"])
    ws = _workspace(mlp, "fused_bwd", int(lib.nm_mlp_backward_fused_workspace_bytes(mlp.handle)))
    ct = _tape_struct(tape)
    with _stage("delta"):        # delta chain + all weight gradients, the heads included: one launch + one reduction
        check(lib.nm_mlp_backward_fused(mlp.handle, n, C.byref(ct), _ptr(radiance), _ptr(grad_radiance), None, C.byref(pg),
                                        _ptr(ws), _stream()), "nm_mlp_backward_fused")
    return g


class _MLPRays(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mlp, names, origins, dirs, t, *params):
        radiance, tape = forward_train(mlp, origins, dirs, t)
        ctx.mlp, ctx.names, ctx.tape = mlp, names, tape
        ctx.save_for_backward(origins, dirs, t, radiance)
        return radiance

    @staticmethod
    def backward(ctx, grad):
        origins, dirs, t, radiance = ctx.saved_tensors
        g = backward(ctx.mlp, ctx.tape, radiance, grad, origins, dirs, t)
        ctx.tape = None
        return (None, None, None, None, None) + tuple(g[k] for k in ctx.names)


def mlp_rays(module, origins, dirs, t):
    """Differentiable FlexibleNeRFModel.forward over (R,S) ray samples; `module` is the nn.Module mirror
    (nerfmeshes_amd.nerf.models.FlexibleNeRFModel) whose parameters receive the gradients."""
    for name, x in (("ray origins", origins), ("ray directions", dirs), ("sample depths", t)):
        if isinstance(x, torch.Tensor) and x.requires_grad and torch.is_grad_enabled():
            raise NotImplementedError(
                f"the {name} require a gradient: the HIP backward produces parameter gradients only (what NeRFModel.training_step "
                "asks for, model_nerf.py:88-151); detach them, or they would silently receive none")
    mlp = module.hip()          # the packed copy rebuilt from the live parameters (guard_mode() below)
    names = param_names(int(mlp.desc["num_layers"]), bool(mlp.desc.get("use_viewdirs", True)))
    pack = getattr(module, "_pack", None)                  # hip() has just validated its list of (name, Parameter)
    params = dict(zip(pack[1], pack[2])) if pack else dict(module.named_parameters())
    return _MLPRays.apply(mlp, names, origins, dirs, t, *[params[k] for k in names])


# ---- "the parameters may have changed".  The reference's forward reads the nn.Parameter storages themselves
# (/root/reference/src/nerf/models.py:60-80); a HipMLP handle holds a packed image of them, which FlexibleNeRFModel.hip() keeps
# current under one of three policies (NERFMESHES_WEIGHTS_GUARD, or `module.weights_guard`):
#   "always" (default)  re-pack on EVERY use: one gather kernel on the stream (~5 us, no host round trip) in front of the launch --
#                       whatever edited the tensors (an optimizer, `p.data.mul_()`, torch._foreach_* on `.data`, a replayed
#                       graph, another library), the kernels see it, exactly as the reference's forward would;
#   "key"               re-pack only when the host can tell: autograd's version counters or storage pointers moved, an optimizer
#                       that holds one of the module's parameters stepped (the post-step hook below: torch's FUSED optimizers
#                       move no version counter), a GraphedStep was replayed.  Edits through `p.data` are invisible here: call
#                       FlexibleNeRFModel.refresh() after them.  For launch-bound loops of tiny batches;
#   "check"             "key", plus a device checksum of the live tensors against the packed image on every use whose key did
#                       not move (nm_mlp_weights_current: synchronises); a mismatch raises StaleWeightsError -- the CI mode that
#                       proves a loop is safe under "key".
_GENERATION = [0]
_OWNERS = {}          # id(parameter) -> weakref of the FlexibleNeRFModel that packs it (filled by FlexibleNeRFModel.hip())


class StaleWeightsError(RuntimeError):
    pass


def guard_mode(module=None):
    mode = getattr(module, "weights_guard", None) or os.environ.get("NERFMESHES_WEIGHTS_GUARD", "always")
    if mode not in ("always", "key", "check"):
        raise ValueError(f"NERFMESHES_WEIGHTS_GUARD={mode!r}: expected always | key | check")
    return mode


def parameters_changed(*_args, **_kw):
    """Every module's packed copy is out of date (a replayed graph stepped an optimizer without running Python)."""
    _GENERATION[0] += 1


def generation():
    return _GENERATION[0]


# how often ANY module registered a parameter (torch calls the hook below from Module.register_parameter / __setattr__): the cheap
# way for FlexibleNeRFModel.hip() to know that its cached list of Parameter objects is still the module's
_REGISTRATIONS = [0]


def registrations():
    return _REGISTRATIONS[0]


def _parameter_registered(module, name, param):
    _REGISTRATIONS[0] += 1


try:
    from torch.nn.modules.module import (register_module_module_registration_hook as _register_module_hook,
                                         register_module_parameter_registration_hook as _register_param_hook)
    _register_param_hook(_parameter_registered)
    _register_module_hook(_parameter_registered)       # a replaced sub-module brings new Parameter objects with it
    PARAMETER_HOOK = True
except ImportError:         # an older torch: no caching of the parameter list (hip() walks the module tree on every use)
    PARAMETER_HOOK = False


def register_owner(module, params):
    import weakref
    ref = weakref.ref(module)
    for p in params:
        _OWNERS[id(p)] = ref


def _optimizer_stepped(optimizer, *_args, **_kw):
    """Post-step hook: the modules whose parameters THIS optimizer holds are marked changed; optimizers of unrelated modules
    (a discriminator, an EMA copy, another library's model in the same process) touch nothing."""
    seen = set()
    for group in optimizer.param_groups:
        for p in group["params"]:
            ref = _OWNERS.get(id(p))
            if ref is None:
                continue
            module = ref()
            if module is None:
                del _OWNERS[id(p)]
            elif id(module) not in seen:
                seen.add(id(module))
                module._generation = getattr(module, "_generation", 0) + 1


from torch.optim.optimizer import register_optimizer_step_post_hook as _register_step_hook  # noqa: E402

# One process-wide registration of the SCOPED hook above: also optimizers the caller built himself (torch.optim.Adam(...,
# fused=True) straight from the reference's model_base.py:159-162) are seen.  NERFMESHES_GLOBAL_STEP_HOOK=0 leaves torch's
# global hook list alone; make_optimizer() then registers the hook on the optimizers it builds.
_GLOBAL_HOOK = os.environ.get("NERFMESHES_GLOBAL_STEP_HOOK", "1") != "0"
if _GLOBAL_HOOK:
    _register_step_hook(_optimizer_stepped)


def weights_differ(mlp, params, struct=None):
    """True when the packed image of `mlp` no longer equals the live tensors `params` (name -> CUDA fp32 tensor)."""
    w, keep = struct if struct is not None else _weights_struct(mlp, params)
    differs = C.c_int32(0)
    check(_lib.load().nm_mlp_weights_current(mlp.handle, C.byref(w), _stream(), C.byref(differs)), "nm_mlp_weights_current")
    del keep
    return bool(differs.value)


def make_optimizer(kind, params, lr, **kw):
    """torch.optim.<kind>(params, lr=lr) as the reference builds it (model_base.py:159-162) -- in torch's FUSED implementation
    when the class has one and every parameter lives on the GPU: one kernel per step instead of seven multi-tensor launches, the
    same update formula.  Nothing for an 8x256 pair (14.06 vs 14.08 ms per iteration), 4 -- 13 % of an iteration of BASELINE config
    1's 4x64 network (1.27 -- 1.30 -> 1.13 -- 1.22 ms: profiles/r05_train_graph_and_adam.json).  (A fused step moves no autograd
    version counter: FlexibleNeRFModel.hip() re-packs on `generation()` above.)"""
    import inspect
    cls = getattr(torch.optim, kind)
    params = list(params)
    flat = [p for g in params for p in g["params"]] if params and isinstance(params[0], dict) else params
    # (only the optimizers whose fused implementation is a GPU one: torch's Adagrad, e.g., takes the flag for host tensors only)
    if kind in ("Adam", "AdamW", "SGD") and "fused" in inspect.signature(cls.__init__).parameters and "fused" not in kw and \
            "foreach" not in kw and flat and all(p.is_cuda and torch.is_floating_point(p) for p in flat):
        kw["fused"] = True
    opt = cls(params, lr=lr, **kw)
    if not _GLOBAL_HOOK:
        opt.register_step_post_hook(_optimizer_stepped)
    return opt


class GraphedStep:
    """One optimizer iteration of FIXED shapes captured into a hipGraph once, then replayed: for the small networks, whose
    iteration is about forty launches of a few tens of microseconds, the launch gaps are a third of the wall time (BASELINE config
    1's 4x64 network on 8192 rays x 32 samples: 1.27 ms eagerly, 1.52 with the capturable Adam a capture needs, 0.89 replayed --
    9.2 M rays/s; an 8x256 pair gains nothing, its CPU already runs ahead of 2 -- 4 ms kernels: profiles/r05_train_graph_and_adam.json).  The library's training kernels capture as they are: every buffer
    they use comes from the caller, no call allocates, copies from the host or synchronises once the handles are warm.

        step = GraphedStep(iteration)     # iteration(): optimizer.zero_grad(set_to_none=True); forward on STATIC input tensors;
        for batch in loader:              #              loss.backward(); optimizer.step()  -- optimizer built with capturable=True
            static_rays.copy_(batch); step()

    `warmup` eager iterations run first on a side stream (they ARE optimizer steps: allocator, handle re-pack and the cached ray
    bounds must be warm before the capture); the capture itself records the iteration without running it.  The parameters after
    k replays equal those after k eager iterations bit for bit (tests/test_gpu_train.py)."""

    def __init__(self, iteration, warmup=3):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                iteration()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            iteration()

    def __call__(self):
        self.graph.replay()
        parameters_changed()       # the step inside the graph ran without Python: FlexibleNeRFModel.hip() must re-pack on its next use


def perturb_intervals(t, rnd):
    t, rnd = _dev32(t), _dev32(rnd)
    out = torch.empty_like(t)
    check(_lib.load().nm_perturb_intervals(_ptr(t), _ptr(rnd), t.shape[0], t.shape[1], _ptr(out), _stream()),
          "nm_perturb_intervals")
    return out


def sample_pdf_rand(t, weights, u):
    """SamplePDF.forward with per-ray u (R, num_fine) -- modules.py:224-228."""
    t, weights, u = _dev32(t), _dev32(weights), _dev32(u)
    rays, coarse = t.shape
    out = torch.empty(rays, coarse + u.shape[1], dtype=torch.float32, device=t.device)
    check(_lib.load().nm_sample_pdf_rand(_ptr(t), _ptr(weights), _ptr(u), rays, coarse, u.shape[1], _ptr(out), _stream()),
          "nm_sample_pdf_rand")
    return out


class _Composite(torch.autograd.Function):
    @staticmethod
    def forward(ctx, radiance, t, dirs, noise, thr, white_bg):
        rays, samples = t.shape
        tensors, out = hip_ops._alloc_bundle(rays, samples, radiance.device)
        with _stage("compositing_forward"):
            check(_lib.load().nm_composite_train(_ptr(radiance), _ptr(t), _ptr(dirs), _ptr(noise), rays, samples, float(thr),
                                                 int(white_bg), C.byref(out), _stream()), "nm_composite_train")
        ctx.white_bg = bool(white_bg)
        ctx.noise = noise
        ctx.save_for_backward(radiance, t, dirs)
        ctx.set_materialize_grads(False)
        res = tuple(tensors[k] for k in hip_ops.BUNDLE_FIELDS)
        ctx.mark_non_differentiable(tensors["mask_weights"], tensors["disp_map"])
        return res

    @staticmethod
    def backward(ctx, g_rgb, g_depth, g_weights, g_mask, g_acc, g_disp):
        radiance, t, dirs = ctx.saved_tensors
        rays, samples = t.shape
        keep = [None if g is None else _dev32(g) for g in (g_rgb, g_acc, g_depth, g_weights)]
        grads = BundleGrads(*[_ptr(g) for g in keep])
        out = torch.empty_like(radiance)
        with _stage("compositing_backward"):
            check(_lib.load().nm_composite_backward(_ptr(radiance), _ptr(t), _ptr(dirs), _ptr(ctx.noise), rays, samples,
                                                    int(ctx.white_bg), C.byref(grads), _ptr(out), _stream()),
                  "nm_composite_backward")
        return out, None, None, None, None, None


def composite(radiance, t, dirs, noise=None, attenuation_threshold=1e-5, white_background=False):
    """Differentiable VolumeRenderer.forward in training mode -> dict of the six bundle fields."""
    for name, x in (("sample depths", t), ("ray directions", dirs)):
        if isinstance(x, torch.Tensor) and x.requires_grad and torch.is_grad_enabled():
            raise NotImplementedError(f"the {name} require a gradient: the compositing backward differentiates the radiance only "
                                      "(modules.py:67-121 as NeRFModel.training_step uses it); detach them")
    radiance = radiance if radiance.is_contiguous() else radiance.contiguous()
    t, dirs = _dev32(t), _dev32(dirs, radiance.device)
    noise = None if noise is None else _dev32(noise, radiance.device)
    res = _Composite.apply(radiance, t, dirs, noise, attenuation_threshold, white_background)
    return dict(zip(hip_ops.BUNDLE_FIELDS, res))
