"""Runs the UNMODIFIED reference scripts (`/root/reference/src/{eval_nerf,mesh_nerf,train_nerf}.py`) as `__main__`
against `nerfmeshes_amd.compat.install()` in a fresh interpreter (tests/test_reference_scripts.py launches it; the
shim's module names `models` / `nerf` / `data` must not leak into the pytest process, where other tests import the real
reference under the same names).

Two backends (`NM_REF_BACKEND`):

* `hip` -- the REAL `models.NeRFModel` over the HIP kernels on cuda:0: one process in which the unmodified
  `/root/reference/src/eval_nerf.py:62-65 -> models.NeRFModel.query -> nm_render_rays` (and `mesh_nerf.py:73-79 ->
  sample_points -> nm_mlp_sample_points`, `skimage.measure.marching_cubes -> nm_mc_count / nm_mc_emit`, `train_nerf.py ->
  training_step -> nm_mlp_forward_train / nm_mlp_backward`) actually executes.  Needs a GPU AND a reference checkout
  (`NERFMESHES_REFERENCE=<dir>`); the expectations still come from the oracle, on the CPU.
* `oracle` (default) -- there is no GPU in the build container and no reference tree on the driver's GPU box, so here the
  arithmetic behind the shim's model classes is replaced by the CPU ORACLE (test double `OracleNeRFModel`, registered as
  `models.OracleNeRFModel`):
what is under test is everything between the reference's script and the kernels -- module aliases, third-party
stand-ins, PathParser, dataset classes, DataLoader collation, DataBundle, batchify, the cast_* / export_obj helpers,
BaseModel.setup / dataloaders, the Trainer stand-in, LoggerCallback, checkpoint layout.  The GPU tests
(tests/test_gpu_reference_flow.py) run the package's mirrors of the same scripts on the real kernels and compare with
the same oracle bookkeeping.

    python tests/tools/ref_script_runner.py <scenario> <workdir>      -> last stdout line: JSON
"""
import contextlib
import io
import json
import os
import runpy
import sys

import numpy as np
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF_SRC = os.environ.get("NERFMESHES_REFERENCE", "/root/reference") + "/src"

from nerfmeshes_amd import compat, synthetic as S  # noqa: E402
from oracle import mc_oracle, nerf_oracle as O      # noqa: E402  (test double + checker)

models, nerf = compat.install()
from nerfmeshes_amd.data import CachedRayDataset, DataBundle, DatasetType  # noqa: E402
from nerfmeshes_amd.nerf.modules import OutputBundle                        # noqa: E402

BACKEND = os.environ.get("NM_REF_BACKEND", "oracle")
HIP = BACKEND == "hip"
if HIP and not torch.cuda.is_available():
    raise SystemExit("NM_REF_BACKEND=hip needs a MI355X")
DEVICE = "cuda" if HIP else "cpu"
MLP = dict(num_layers=4, hidden_size=32, skip_step=2, num_encoding_fn_xyz=4, num_encoding_fn_dir=2)
MODEL_NAME = "NeRFModel" if HIP else "OracleNeRFModel"      # the class the checkpoints name, i.e. what the scripts instantiate
H, W, FOCAL = 10, 14, 16.0


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


def hparams(work, **over):
    hp = S.hparams(model=MODEL_NAME, num_coarse=8, num_fine=8, chunksize=48, **MLP)
    hp.update({"experiment.logdir": os.path.join(work, "logs"), "dataset.caching.use_caching": True,
               "dataset.caching.cache_dir": os.path.join(work, "cache"), "nerf.train.num_random_rays": 64,
               "nerf.train.chunksize": 48, "experiment.train_iters": 6, "experiment.validate_every": 3,
               "experiment.print_every": 2, "optimizer.lr": 1e-2})
    hp.update(over)
    return hp


def write_cache(hp, counts=(3, 2, 2)):
    """Ray caches for the three splits: rays from the oracle's get_ray_bundle, seeded pseudo-photographs as targets."""
    cfg = nerf.CfgNode(models.nest_dict(hp, sep="."))
    g = torch.Generator().manual_seed(5)
    poses = S.orbit_poses(sum(counts))
    k = 0
    for split, n in zip((DatasetType.TRAIN, DatasetType.VALIDATION, DatasetType.TEST), counts):
        ds = CachedRayDataset(cfg, split)
        for i in range(n):
            o, d = O.get_ray_bundle(H, W, FOCAL, torch.as_tensor(poses[k]))
            ds.write_view(DataBundle(ray_origins=o, ray_directions=d, ray_targets=torch.rand(H, W, 3, generator=g),
                                     ray_bounds=torch.tensor([2.0, 6.0]), hwf=(H, W, FOCAL), size=1), i)
            k += 1
    return cfg


def write_checkpoint(hp, seed=3):
    """`<logdir>/<exp>/default/version_0/{hparams.yaml, checkpoints/model_last.ckpt}` with seeded weights."""
    torch.manual_seed(seed)
    m = OracleNeRFModel(dict(hp))
    with torch.no_grad():
        for net in (m.model_coarse, m.model_fine):
            net.fc_alpha.weight.mul_(40.0)
    vdir = os.path.join(hp["experiment.logdir"], hp["experiment.id"], "default", "version_0")
    os.makedirs(os.path.join(vdir, "checkpoints"), exist_ok=True)
    with open(os.path.join(vdir, "hparams.yaml"), "w") as fh:
        yaml.safe_dump(dict(hp), fh)
    m.save_checkpoint(os.path.join(vdir, "checkpoints", "model_last.ckpt"))
    return vdir, m


def run_reference(script, argv):
    """Execute a reference script as __main__ with `argv`; returns its stdout."""
    old, buf = sys.argv, io.StringIO()
    sys.argv = [os.path.join(REF_SRC, script)] + argv
    try:
        with contextlib.redirect_stdout(buf):
            runpy.run_path(os.path.join(REF_SRC, script), run_name="__main__")
    finally:
        sys.argv = old
    return buf.getvalue()


def png(path):
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im)


def scenario_imports(work):
    out = {}
    for s in ("eval_nerf", "mesh_nerf", "train_nerf"):
        ns = runpy.run_path(os.path.join(REF_SRC, s + ".py"), run_name="reference_" + s)
        out[s] = sorted(k for k in ns if not k.startswith("_") and callable(ns[k]))
    out["stand_ins"] = sorted(n for n, m in sys.modules.items() if getattr(m, "__nerfmeshes_amd_stand_in__", None) is True)
    return out


def scenario_eval(work):
    hp = hparams(work)
    cfg = write_cache(hp)
    vdir, model = write_checkpoint(hp)
    save_ref, save_ours = os.path.join(work, "out_ref"), os.path.join(work, "out_ours")
    text = run_reference("eval_nerf.py", ["--log-checkpoint", vdir, "--save-dir", save_ref, "--save-images", "--save-disparity"])
    # expectation straight from the oracle's bookkeeping (R8)
    test = CachedRayDataset(cfg, DatasetType.TEST)
    model.eval()
    losses, rgbs = [], []
    with torch.no_grad():
        for i in range(len(test)):
            b = DataBundle.deserialize(test[i]).to_ray_batch()
            rgb = torch.cat([model.query((b.ray_origins, b.ray_directions[s:s + 48], b.ray_bounds)).rgb_map
                             for s in range(0, H * W, 48)])
            rgbs.append(rgb)
            losses.append(float(O.view_loss(rgb, b.ray_targets, 48)))
        # the package's mirror of the script, same model object, same files
        from nerfmeshes_amd import eval_nerf as mirror
        args = mirror.build_parser().parse_args(["--log-checkpoint", vdir, "--save-dir", save_ours, "--save-images", "--save-disparity"])
        run_model = model
        if HIP:     # the mirror over the real kernels too: script and mirror must agree bit for bit, the oracle within tolerance
            run_model = models.NeRFModel.load_from_checkpoint(os.path.join(vdir, "checkpoints", "model_last.ckpt")).eval().to(DEVICE)
        with contextlib.redirect_stdout(io.StringIO()) as mtext:
            total = mirror.eval_nerf(run_model, args, cfg, DEVICE)
    files = sorted(os.path.relpath(os.path.join(d, f), save_ref) for d, _, fs in os.walk(save_ref) for f in fs)
    same = all(np.array_equal(png(os.path.join(save_ref, f)), png(os.path.join(save_ours, f))) for f in files)
    img0 = png(os.path.join(save_ref, hp["experiment.id"], "images", "0000.png"))
    want0 = (rgbs[0].view(H, W, 3).clamp(0, 1) * 255).to(torch.uint8).numpy()
    return {"backend": BACKEND, "stdout": text, "expected_losses": losses, "expected_total": float(O.dataset_loss(losses)),
            "mirror_total": float(total), "mirror_stdout": mtext.getvalue(), "files": files,
            "mirror_files_identical": bool(same), "image0_matches_render": bool(np.array_equal(img0, want0))}


def scenario_mesh(work):
    import skimage.measure
    if not HIP:      # hip: whatever `skimage.measure.marching_cubes` resolves to -- the package, or compat's nm_mc_* stand-in
        skimage.measure.marching_cubes = lambda vol, level: mc_oracle.marching_cubes(np.ascontiguousarray(vol), float(level))
    hp = hparams(work)
    vdir, model = write_checkpoint(hp)
    res = 20
    out = {}
    for tag, extra in (("view", ["--view-disparity-max-bound", "1.0"]), ("diffuse", ["--no-view-dependence"])):
        save = os.path.join(work, "mesh_" + tag)
        os.makedirs(save, exist_ok=True)
        text = run_reference("mesh_nerf.py", ["--log-checkpoint", vdir, "--res", str(res), "--iso-level", "5", "--limit", "1.2",
                                              "--save-dir", save, "--batch-size", "3000", "--override-cache-mesh"] + extra)
        lines = open(os.path.join(save, "mesh.obj")).read().splitlines()
        out[tag] = {"stdout": text, "v": sum(l.startswith("v ") for l in lines), "vn": sum(l.startswith("vn ") for l in lines),
                    "f": sum(l.startswith("f ") for l in lines), "first_v": lines[0], "cache": os.path.exists(os.path.join(save, "mesh_cache.pt"))}
    # expectation: oracle grid -> numpy iso clamp -> C oracle marching cubes
    with torch.no_grad():
        pts = O.grid_points(1.2, res)
        sigma = model.sample_points(pts, pts)[:, 3].reshape(res, res, res).numpy()
    iso = min(max(5.0, sigma.min() + sigma.std()), sigma.max() - sigma.std())
    v, f, n, _ = mc_oracle.marching_cubes(sigma, float(iso))
    out["expected"] = {"v": int(v.shape[0]), "f": int(f.shape[0]), "iso": float(iso)}
    out["backend"] = BACKEND
    out["marching_cubes_is_stand_in"] = bool(getattr(skimage.measure, "__nerfmeshes_amd_stand_in__", False))
    return out


def scenario_train(work):
    hp = hparams(work)
    write_cache(hp)
    cfg_path = os.path.join(work, "experiment.yml")
    with open(cfg_path, "w") as fh:
        yaml.safe_dump(models.nest_dict(hp, sep="."), fh)           # nested, like config/*.yml
    text = run_reference("train_nerf.py", ["--config", cfg_path, "--run-name", "unit", "--deterministic"])
    vdir = os.path.join(hp["experiment.logdir"], hp["experiment.id"], "unit", "version_0")
    ck = os.path.join(vdir, "checkpoints", "model_last.ckpt")
    state = torch.load(ck, weights_only=False)
    metrics = [json.loads(l) for l in open(os.path.join(vdir, "metrics.jsonl"))]
    # resume from the log directory for 4 more steps (train_nerf.py --log-checkpoint)
    flat = yaml.safe_load(open(os.path.join(vdir, "hparams.yaml")))
    flat["experiment.train_iters"] = 10
    with open(os.path.join(vdir, "hparams.yaml"), "w") as fh:
        yaml.safe_dump(flat, fh)
    text2 = run_reference("train_nerf.py", ["--log-checkpoint", vdir])
    state2 = torch.load(ck, weights_only=False)
    reloaded = OracleNeRFModel.load_from_checkpoint(ck)
    return {"backend": BACKEND, "stdout": text, "resume_stdout": text2, "checkpoint_keys": sorted(state.keys()),
            "checkpoints": sorted(os.listdir(os.path.join(vdir, "checkpoints"))),
            "global_step": int(state["global_step"]), "resumed_global_step": int(state2["global_step"]),
            "hparams_yaml": os.path.exists(os.path.join(vdir, "hparams.yaml")),
            "train_losses": [m["train/loss"] for m in metrics if "train/loss" in m],
            "state_dict_keys": len(state["state_dict"]), "reloaded_params": sum(p.numel() for p in reloaded.parameters()),
            "weights_moved": bool(any(not torch.equal(state["state_dict"][k], state2["state_dict"][k]) for k in state["state_dict"]))}


if __name__ == "__main__":
    name, work = sys.argv[1], sys.argv[2]
    os.makedirs(work, exist_ok=True)
    result = {"imports": scenario_imports, "eval": scenario_eval, "mesh": scenario_mesh, "train": scenario_train}[name](work)
    if HIP:
        # evidence that the kernels ran in THIS process: the library is mapped and its MLP launch counter moved
        from nerfmeshes_amd import _lib
        result["native_library"] = os.path.basename(_lib.load()._name)
    print(json.dumps(result))
