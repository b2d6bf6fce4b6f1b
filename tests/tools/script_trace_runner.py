"""The drop-in claim as a CALL TRACE: what the reference's three scripts ask of the model API, recorded here from the UNMODIFIED
scripts and replayed on the MI355X by the package's command lines, at the shapes the reference ships.

The reference checkout cannot travel to the GPU box in any form, and the build container has no GPU: the unmodified
`/root/reference/src/{eval_nerf,mesh_nerf,train_nerf}.py` can therefore never run in ONE process with the real kernels.  What can
be shown instead, and is:

  container (here)   reference script as __main__  over compat.install(), arithmetic = the CPU oracle (test double)  -> TRACE_ref
  MI355X box         package command line (mirror)  over the SAME shim classes, arithmetic = the HIP kernels          -> TRACE_hip

`trace` = every call that crosses the model API -- `load_from_checkpoint`, `query`, `sample_points`, `forward` (training),
`training_step`, `validation_step`, `configure_optimizers`, `skimage.measure.marching_cubes`, `export_obj` -- with the shapes /
dtypes of its tensor arguments and its scalar arguments, run-length encoded.  TRACE_ref is committed as a fixture
(tests/golden/script_traces.json, written by tests/golden/make_script_traces.py together with the numbers the reference's script
PRINTED); tests/test_gpu_script_traces.py asserts TRACE_hip == TRACE_ref entry for entry and the printed numbers within the
render tolerance.  Everything between the script's own lines and the kernels (shim classes, DataBundle, batchify, Trainer
stand-in, LoggerCallback, checkpoint layout) is the same code in both runs; the script's own lines are the only thing the GPU
run replaces, and the trace is the proof that the replacement asks the kernels for the same work.

    NM_REF_BACKEND=oracle|hip  NM_REF_WHICH=reference|mirror  NM_REF_SHAPES=tiny|shipped
    python tests/tools/script_trace_runner.py <eval|mesh|train> <workdir>      -> last stdout line: JSON
"""
import contextlib
import io
import json
import os
import re
import runpy
import sys
import time

import numpy as np
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF_SRC = os.environ.get("NERFMESHES_REFERENCE", "/root/reference") + "/src"
BACKEND = os.environ.get("NM_REF_BACKEND", "oracle")
WHICH = os.environ.get("NM_REF_WHICH", "reference")
SHAPES = os.environ.get("NM_REF_SHAPES", "tiny")
HIP = BACKEND == "hip"
if HIP and not torch.cuda.is_available():
    raise SystemExit("NM_REF_BACKEND=hip needs a MI355X")
if WHICH == "mirror" and not HIP:
    raise SystemExit("the package's command lines have no CPU path: NM_REF_WHICH=mirror needs NM_REF_BACKEND=hip")
DEVICE = "cuda" if HIP else "cpu"

from nerfmeshes_amd import compat, synthetic as S  # noqa: E402
from oracle import mc_oracle, nerf_oracle as O      # noqa: E402  (test double of the CPU run + checker)

models, nerf = compat.install()
from nerfmeshes_amd.data import CachedRayDataset, DataBundle, DatasetType  # noqa: E402
from nerfmeshes_amd.nerf.modules import OutputBundle                        # noqa: E402

# ---- the two sets of shapes -------------------------------------------------------------------------------------------------
#   shipped: eval / mesh on the headline network (8x256, 64 + 128 samples, chunks of 2048: config/nerf-synthetic-lego, BASELINE
#            configs 2 - 4), training on nerf-colmap-fern's 8x128 (/root/reference/config/nerf-colmap-fern.yml:115,152)
#   tiny:    the 4x32 network of tests/tools/ref_script_runner.py (seconds on the CPU: the CPU suite's cross-check)
MLP_256 = dict(num_layers=8, hidden_size=256, skip_step=4, num_encoding_fn_xyz=10, num_encoding_fn_dir=4)
MLP_128 = dict(num_layers=8, hidden_size=128, skip_step=4, num_encoding_fn_xyz=10, num_encoding_fn_dir=4)
MLP_TINY = dict(num_layers=4, hidden_size=32, skip_step=2, num_encoding_fn_xyz=4, num_encoding_fn_dir=2)
CFG = {
    "shipped": dict(eval=dict(mlp=MLP_256, coarse=64, fine=128, chunk=2048, size=(100, 100), views=(1, 1, 2)),
                    mesh=dict(mlp=MLP_256, coarse=64, fine=128, chunk=2048, res=128, iso=32.0, limit=1.2, batch=None),
                    train=dict(mlp=MLP_128, coarse=64, fine=128, chunk=2048, size=(40, 40), views=(3, 2, 2), rays=1024, iters=20,
                               validate_every=10, print_every=5, resume_to=24, lr=5e-4)),
    "tiny": dict(eval=dict(mlp=MLP_TINY, coarse=8, fine=8, chunk=48, size=(10, 14), views=(3, 2, 2)),
                 mesh=dict(mlp=MLP_TINY, coarse=8, fine=8, chunk=48, res=20, iso=5.0, limit=1.2, batch=3000),
                 train=dict(mlp=MLP_TINY, coarse=8, fine=8, chunk=48, size=(10, 14), views=(3, 2, 2), rays=64, iters=6,
                            validate_every=3, print_every=2, resume_to=10, lr=1e-2)),
}[SHAPES]
FOCAL = {"shipped": 138.9, "tiny": 16.0}[SHAPES]


class OracleNeRFModel(models.NeRFModel):
    """`models.NeRFModel` with the CPU oracle in place of the HIP kernels (differentiable: plain torch ops)."""

    def _spec(self, part):
        c = self.cfg.models[part]
        return O.MLPSpec(num_layers=c.num_layers, hidden_size=c.hidden_size, skip_step=c.skip_step,
                         num_encoding_fn_xyz=c.num_encoding_fn_xyz, num_encoding_fn_dir=c.num_encoding_fn_dir)

    def forward(self, x):
        origins, dirs, (near, far) = x
        nerf_cfg = self.cfg.nerf.train if self.model_coarse.training else self.cfg.nerf.validation
        rs = O.RenderSpec(num_coarse=self.sampler.count, num_fine=self.sample_pdf.num_samples, lindisp=nerf_cfg.lindisp,
                          white_background=self.cfg.dataset.white_background)
        rays = dirs.shape[0]
        t = O.coarse_intervals(float(near), float(far), rs.num_coarse, rays, rs.lindisp)

        def run(net, part, t):
            pts = O.ray_points(t, dirs, origins)
            rad = O.mlp_forward(dict(net.named_parameters()), self._spec(part), pts.reshape(-1, 3),
                                dirs[:, None, :].expand_as(pts).reshape(-1, 3), keep_graph=True).reshape(rays, -1, 4)
            out = O.composite(rad, t, dirs, rs)
            return OutputBundle(**{k: out[k] for k in ("rgb_map", "depth_map", "weights", "mask_weights", "acc_map", "disp_map")})

        coarse = run(self.model_coarse, "coarse", t)
        if self.model_fine is None:
            return coarse, None
        tf = O.sample_pdf_intervals(t, coarse.weights.detach(), rs.num_fine)
        return coarse, run(self.model_fine, "fine", tf)

    def sample_points(self, points, rays=None, **kwargs):
        net = self.get_model()
        part = "fine" if self.model_fine is not None else "coarse"
        return O.mlp_forward(dict(net.named_parameters()), self._spec(part), points, rays if rays is not None else points,
                             keep_graph=True)


models.OracleNeRFModel = OracleNeRFModel
MODEL_NAME = "NeRFModel" if HIP else "OracleNeRFModel"
MODEL_CLS = getattr(models, MODEL_NAME)


# ---- the recorder ------------------------------------------------------------------------------------------------------------
def _describe(x, depth=0):
    if isinstance(x, torch.Tensor):
        return ["T", list(x.shape), str(x.dtype).replace("torch.", "")]
    if isinstance(x, np.ndarray):
        return ["A", list(x.shape), str(x.dtype)]
    if isinstance(x, (bool, int, str)) or x is None:
        return x
    if isinstance(x, float):
        return float(f"{x:.5g}")
    if isinstance(x, (np.floating, np.integer)):
        return _describe(x.item())
    if isinstance(x, (tuple, list)) and depth < 3:
        return [_describe(v, depth + 1) for v in x]
    if isinstance(x, DataBundle):
        return {"DataBundle": {k: _describe(v, depth + 1) for k, v in sorted(vars(x).items()) if isinstance(v, torch.Tensor)}}
    if isinstance(x, dict) and depth < 3:
        return {str(k): _describe(v, depth + 1) for k, v in sorted(x.items(), key=lambda kv: str(kv[0]))
                if isinstance(v, (torch.Tensor, np.ndarray, bool, int, float, str))}
    return type(x).__name__


class Recorder:
    def __init__(self):
        self.events, self.depth = [], 0

    def log(self, name, args, kwargs, describe=None):
        if self.depth:                       # only calls that cross the API from outside: what the SCRIPT asks for
            return
        describe = describe or _describe
        ev = [name, [describe(a) for a in args], {k: describe(v) for k, v in sorted(kwargs.items())}]
        if self.events and self.events[-1][0] == ev:
            self.events[-1][1] += 1
        else:
            self.events.append([ev, 1])

    def wrap_method(self, cls, name, label=None, skip_self=True):
        orig = getattr(cls, name)
        rec, label = self, label or name

        def wrapper(*args, **kwargs):
            rec.log(label, args[1:] if skip_self else args, kwargs)
            rec.depth += 1
            try:
                return orig(*args, **kwargs)
            finally:
                rec.depth -= 1
        wrapper.__wrapped__ = orig
        setattr(cls, name, wrapper)

    def wrap_function(self, original, label, transparent=False, describe=None):
        """Replace `original` wherever a loaded module holds it as an attribute (the scripts bind names at import time, the
        package's modules did so long ago).  `transparent`: calls made inside it still count as the script's own."""
        rec = self

        def wrapper(*args, **kwargs):
            rec.log(label, args, kwargs, describe)
            if transparent:
                return original(*args, **kwargs)
            rec.depth += 1
            try:
                return original(*args, **kwargs)
            finally:
                rec.depth -= 1
        wrapper.__wrapped__ = original
        for mod in list(sys.modules.values()):
            for attr, val in list(getattr(mod, "__dict__", {}).items()):
                if val is original:
                    setattr(mod, attr, wrapper)
        return wrapper


REC = Recorder()
LOADED = []           # every model a command line loaded from a checkpoint (kept alive: their handles' re-pack counters are read at the end)


def install_recorder():
    for name in ("query", "sample_points", "forward", "training_step", "validation_step", "configure_optimizers"):
        # resolved on the concrete class the checkpoints name (the oracle double overrides forward / sample_points)
        REC.wrap_method(MODEL_CLS, name)
    orig = MODEL_CLS.load_from_checkpoint.__func__

    def load_from_checkpoint(cls, path, *a, **k):
        REC.log("load_from_checkpoint", (os.path.basename(str(path)),), {})
        REC.depth += 1
        try:
            model = orig(cls, path, *a, **k)
            LOADED.append(model)
            return model
        finally:
            REC.depth -= 1
    MODEL_CLS.load_from_checkpoint = classmethod(load_from_checkpoint)
    import skimage.measure
    if not HIP:      # the CPU run's arithmetic: the C oracle behind scikit-image's entry point (the stand-in is a GPU function)
        skimage.measure.marching_cubes = lambda vol, level: mc_oracle.marching_cubes(np.ascontiguousarray(vol), float(level))

    def loose(x):    # data-dependent values (the iso level, vertex / face counts) are compared as NUMBERS, not as trace entries
        if isinstance(x, (float, np.floating)):
            return "float"
        if isinstance(x, (torch.Tensor, np.ndarray)) and x.ndim == 2:
            return [type(x).__name__, ["n", x.shape[1]], str(x.dtype).replace("torch.", "")]
        return _describe(x) if not isinstance(x, str) else "str"
    REC.wrap_function(skimage.measure.marching_cubes, "marching_cubes", describe=loose)
    from nerfmeshes_amd.nerf import nerf_helpers
    REC.wrap_function(nerf_helpers.export_obj, "export_obj", describe=loose)


# ---- fixtures on disk: seeded weights, ray caches, the Lightning log layout --------------------------------------------------
def hparams(work, c, **over):
    hp = S.hparams(model=MODEL_NAME, num_coarse=c["coarse"], num_fine=c["fine"], chunksize=c["chunk"], **c["mlp"])
    hp.update({"experiment.logdir": os.path.join(work, "logs"), "dataset.caching.use_caching": True,
               "dataset.caching.cache_dir": os.path.join(work, "cache")})
    hp.update(over)
    return hp


def write_cache(hp, c):
    """Ray caches for the three splits: rays from the oracle's get_ray_bundle, seeded pseudo-photographs as targets."""
    cfg = nerf.CfgNode(models.nest_dict(hp, sep="."))
    g = torch.Generator().manual_seed(5)
    h, w = c["size"]
    poses = S.orbit_poses(sum(c["views"]))
    k = 0
    for split, n in zip((DatasetType.TRAIN, DatasetType.VALIDATION, DatasetType.TEST), c["views"]):
        ds = CachedRayDataset(cfg, split)
        for i in range(n):
            o, d = O.get_ray_bundle(h, w, FOCAL, torch.as_tensor(poses[k]))
            ds.write_view(DataBundle(ray_origins=o, ray_directions=d, ray_targets=torch.rand(h, w, 3, generator=g),
                                     ray_bounds=torch.tensor([2.0, 6.0]), hwf=(h, w, FOCAL), size=1), i)
            k += 1
    return cfg


def scene_state(c):
    """Seeded weights with visible structure for both networks of the pair: the benchmark scene for the shipped 8x256, a scaled
    seeded draw otherwise."""
    mlp = c["mlp"]
    if mlp["hidden_size"] == 256:
        w = S.make_scene_weights(**mlp)
    else:
        w = S.make_mlp_weights(3, density_gain=40.0, density_bias=0.3, **mlp)
    return {f"{part}.{k}": torch.as_tensor(np.asarray(v), dtype=torch.float32) for part in ("model_coarse", "model_fine") for k, v in w.items()}


def write_checkpoint(hp, c):
    """`<logdir>/<exp>/default/version_0/{hparams.yaml, checkpoints/model_last.ckpt}` with the seeded scene."""
    torch.manual_seed(3)
    m = OracleNeRFModel(dict(hp))
    sd = m.state_dict()
    sd.update({k: v for k, v in scene_state(c).items() if k in sd})
    m.load_state_dict(sd)
    vdir = os.path.join(hp["experiment.logdir"], hp["experiment.id"], "default", "version_0")
    os.makedirs(os.path.join(vdir, "checkpoints"), exist_ok=True)
    with open(os.path.join(vdir, "hparams.yaml"), "w") as fh:
        yaml.safe_dump(dict(hp), fh)
    m.save_checkpoint(os.path.join(vdir, "checkpoints", "model_last.ckpt"))
    return vdir, m


def run_cli(script, argv):
    """The reference's script as __main__, or the package's command line of the same name; returns (stdout, seconds)."""
    buf, t0 = io.StringIO(), time.perf_counter()
    if WHICH == "reference":
        old = sys.argv
        sys.argv = [os.path.join(REF_SRC, script + ".py")] + argv
        try:
            with contextlib.redirect_stdout(buf):
                runpy.run_path(os.path.join(REF_SRC, script + ".py"), run_name="__main__")
        finally:
            sys.argv = old
    else:
        import importlib
        mirror = importlib.import_module("nerfmeshes_amd." + script)
        with contextlib.redirect_stdout(buf):
            mirror.main(argv)
    if HIP:
        torch.cuda.synchronize()
    return buf.getvalue(), time.perf_counter() - t0


NUM = r"([-+0-9.eE]+)"


def _floats(text, pattern):
    return [float(m.group(1)) for line in text.splitlines() for m in [re.match(pattern, line)] if m]


def _repacks():
    """nm_mlp_refresh_count of the networks of every model a command line loaded (hip backend): [coarse, fine] per model --
    how often the packed parameters were (re)built, the handle's creation included."""
    if not HIP:
        return None
    return [[net.refresh_count() for net in (m.model_coarse, m.model_fine) if net is not None] for m in LOADED]


def scenario_eval(work):
    c = CFG["eval"]
    hp = hparams(work, c)
    write_cache(hp, c)
    vdir, _ = write_checkpoint(hp, c)
    text, dt = run_cli("eval_nerf", ["--log-checkpoint", vdir, "--save-dir", os.path.join(work, "out"), "--save-images", "--save-disparity"])
    return {"stdout_losses": _floats(text, r"\[EVAL\] Iter: \d+ Loss MSE (?:tensor\()?" + NUM),
            "stdout_total": _floats(text, r"Dataset loss MSE: (?:tensor\()?" + NUM),
            "files": sorted(os.path.relpath(os.path.join(d, f), os.path.join(work, "out")) for d, _, fs in os.walk(os.path.join(work, "out")) for f in fs),
            "wall_s": dt, "rays": c["views"][2] * c["size"][0] * c["size"][1]}


def scenario_mesh(work):
    c = CFG["mesh"]
    hp = hparams(work, c)
    vdir, _ = write_checkpoint(hp, c)
    out = {}
    for tag, extra in (("view", ["--view-disparity-max-bound", "1.0"]), ("diffuse", ["--no-view-dependence"])):
        save = os.path.join(work, "mesh_" + tag)
        os.makedirs(save, exist_ok=True)
        argv = ["--log-checkpoint", vdir, "--res", str(c["res"]), "--iso-level", str(c["iso"]), "--limit", str(c["limit"]),
                "--save-dir", save, "--override-cache-mesh"] + extra
        if c["batch"]:
            argv += ["--batch-size", str(c["batch"])]          # shipped shapes: the script's own default (1024) stays
        if WHICH == "mirror":
            argv += ["--route", "script"]
        text, dt = run_cli("mesh_nerf", argv)
        lines = open(os.path.join(save, "mesh.obj")).read().splitlines()
        out[tag] = {"v": sum(l.startswith("v ") for l in lines), "vn": sum(l.startswith("vn ") for l in lines),
                    "f": sum(l.startswith("f ") for l in lines), "first_v": lines[0], "first_f": next(l for l in lines if l.startswith("f ")),
                    "iso": _floats(text, r"Querying based on iso level: " + NUM), "wall_s": dt,
                    "cache": os.path.exists(os.path.join(save, "mesh_cache.pt")), "finished": "Finished writing" in text}
    return out


def scenario_train(work):
    c = CFG["train"]
    hp = hparams(work, c, **{"nerf.train.num_random_rays": c["rays"], "nerf.train.chunksize": c["chunk"],
                             "experiment.train_iters": c["iters"], "experiment.validate_every": c["validate_every"],
                             "experiment.print_every": c["print_every"], "optimizer.lr": c["lr"]})
    write_cache(hp, c)
    cfg_path = os.path.join(work, "experiment.yml")
    with open(cfg_path, "w") as fh:
        yaml.safe_dump(models.nest_dict(hp, sep="."), fh)           # nested, like config/*.yml
    text, dt = run_cli("train_nerf", ["--config", cfg_path, "--run-name", "unit", "--deterministic"])
    vdir = os.path.join(hp["experiment.logdir"], hp["experiment.id"], "unit", "version_0")
    ck = os.path.join(vdir, "checkpoints", "model_last.ckpt")
    state = torch.load(ck, weights_only=False)
    metrics = [json.loads(l) for l in open(os.path.join(vdir, "metrics.jsonl"))]
    flat = yaml.safe_load(open(os.path.join(vdir, "hparams.yaml")))
    flat["experiment.train_iters"] = c["resume_to"]
    with open(os.path.join(vdir, "hparams.yaml"), "w") as fh:
        yaml.safe_dump(flat, fh)
    text2, dt2 = run_cli("train_nerf", ["--log-checkpoint", vdir])
    state2 = torch.load(ck, weights_only=False)
    # ... and the evaluation of what was trained (eval_nerf.py on the resumed checkpoint)
    text3, dt3 = run_cli("eval_nerf", ["--log-checkpoint", vdir, "--save-dir", os.path.join(work, "out")])
    out = {"train_lines": [l for l in text.splitlines() if l.startswith("[TRAIN]") or l.startswith("[VAL]")][:4],
           "done": "Done!" in text, "checkpoints": sorted(os.listdir(os.path.join(vdir, "checkpoints"))),
           "global_step": int(state["global_step"]), "resumed_global_step": int(state2["global_step"]),
           "train_losses": [m["train/loss"] for m in metrics if "train/loss" in m],
           "state_dict_keys": len(state["state_dict"]),
           "weights_moved": bool(any(not torch.equal(state["state_dict"][k], state2["state_dict"][k]) for k in state["state_dict"])),
           "eval_losses": _floats(text3, r"\[EVAL\] Iter: \d+ Loss MSE (?:tensor\()?" + NUM),
           "eval_total": _floats(text3, r"Dataset loss MSE: (?:tensor\()?" + NUM),
           "wall_s": {"train": dt, "resume": dt2, "eval": dt3}}
    if HIP:
        # the checker: the oracle's bookkeeping (R8) on the checkpoint the GPU run trained, views as the script reads them
        REC.depth += 1                       # the checker's own calls are not the command line's
        cfg = nerf.CfgNode(models.nest_dict(flat, sep="."))
        model = OracleNeRFModel.load_from_checkpoint(ck).eval()
        test = CachedRayDataset(cfg, DatasetType.TEST)
        losses = []
        with torch.no_grad():
            for i in range(len(test)):
                b = DataBundle.deserialize(test[i]).to_ray_batch()
                n = b.ray_directions.shape[0]
                rgb = torch.cat([model.query((b.ray_origins, b.ray_directions[s:s + c["chunk"]], b.ray_bounds)).rgb_map for s in range(0, n, c["chunk"])])
                losses.append(float(O.view_loss(rgb, b.ray_targets, c["chunk"])))
        out["oracle_eval_losses_on_this_checkpoint"] = losses
        REC.depth -= 1
        LOADED.pop()
    return out


if __name__ == "__main__":
    name, work = sys.argv[1], sys.argv[2]
    os.makedirs(work, exist_ok=True)
    install_recorder()
    result = {"eval": scenario_eval, "mesh": scenario_mesh, "train": scenario_train}[name](work)
    result.update(backend=BACKEND, which=WHICH, shapes=SHAPES, trace=REC.events, repacks=_repacks(),
                  marching_cubes_is_stand_in=bool(getattr(sys.modules.get("skimage.measure"), "__nerfmeshes_amd_stand_in__", False)))
    if HIP:
        from nerfmeshes_amd import _lib
        result["native_library"] = os.path.basename(_lib.load()._name)
        result["abi_version"] = int(_lib.load().nm_abi_version())
    print(json.dumps(result))
