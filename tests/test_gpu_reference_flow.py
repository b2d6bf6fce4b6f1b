"""GPU: the package's mirrors of the reference's three command lines, end to end on the real kernels, from files on disk.

The reference's scripts themselves cannot run here (no /root/reference on the GPU box); tests/test_reference_scripts.py
executes them in the build container over the same shim with the oracle as arithmetic and shows that the mirrors
(`nerfmeshes_amd.{train,eval,mesh}_nerf`, same flags, same function signatures) reproduce them.  Here the mirrors run
with the HIP path underneath: a NeRF-synthetic scene on disk (transforms_*.json + PNGs) -> `train_nerf --config` (Trainer,
BaseModel.setup, BlenderDataset, GPU ray generation, HIP forward/backward) -> `eval_nerf --log-checkpoint` (PSNR, images)
-> the reference-shaped loop spelled out call by call -> all compared with the CPU oracle on the checkpoint's weights."""
import contextlib
import io
import os

import numpy as np
import pytest
import torch
import yaml

from nerfmeshes_amd import synthetic as S
from oracle import nerf_oracle as O

pytestmark = pytest.mark.gpu
SIZE = 40
MLP = dict(num_layers=4, hidden_size=64, skip_step=2, num_encoding_fn_xyz=6, num_encoding_fn_dir=4)


@pytest.fixture(scope="module")
def scene(tmp_path_factory):
    """Teacher renders of the seeded scene stored in the NeRF-synthetic layout: 4 train / 1 val / 3 test images.  (Not 2:
    `DataBundle.__getitem__(int)` indexes every tensor whose first dimension equals `size`, the (2,) ray bounds included --
    the reference's own quirk, data_helpers.py:94-104.)"""
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a MI355X")
    from nerfmeshes_amd import train_synthetic
    root = tmp_path_factory.mktemp("scene")
    stored = train_synthetic.write_blender_scene(str(root / "lego"), size=SIZE, counts=(4, 1, 3))
    return root, stored


def _config(root, **over):
    hp = S.hparams(num_coarse=32, num_fine=32, chunksize=700, train_perturb=True, train_noise_std=0.0, **MLP)
    hp.update({"experiment.id": "flow", "experiment.logdir": str(root / "logs"), "dataset.basedir": str(root / "lego"),
               "dataset.caching.cache_dir": str(root / "cache"), "nerf.train.num_random_rays": 512,
               "nerf.train.chunksize": 512, "experiment.train_iters": 12, "experiment.validate_every": 8,
               "experiment.print_every": 4, "optimizer.lr": 2e-3, "nerf.validation.num_samples": 1})
    hp.update(over)
    path = root / f"cfg_{abs(hash(tuple(sorted(over.items())))) % 10 ** 8}.yml"
    from nerfmeshes_amd.models import nest_dict
    path.write_text(yaml.safe_dump(nest_dict(hp, sep=".")))
    return str(path), hp


@pytest.fixture(scope="module")
def trained(scene):
    root, _ = scene
    from nerfmeshes_amd import train_nerf
    cfg_path, hp = _config(root)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        trainer, model, pp = train_nerf.main(["--config", cfg_path, "--run-name", "gpu", "--deterministic"])
    return dict(vdir=str(pp.log_dir), stdout=buf.getvalue(), trainer=trainer, hp=hp, root=root)


def _oracle_dataset_loss(ckpt_path, hp, dataset, chunk):
    sd = torch.load(ckpt_path, weights_only=False)["state_dict"]
    wc = {k[len("model_coarse."):]: v for k, v in sd.items() if k.startswith("model_coarse.")}
    wf = {k[len("model_fine."):]: v for k, v in sd.items() if k.startswith("model_fine.")}
    spec, rs = O.MLPSpec(**MLP), O.RenderSpec(num_coarse=32, num_fine=32)
    from nerfmeshes_amd.data import DataBundle
    losses, rgbs = [], []
    for i in range(len(dataset)):
        b = DataBundle.deserialize(dataset[i]).to_ray_batch()
        _, f = O.render(wc, wf, spec, spec, rs, b.ray_origins, b.ray_directions, 2.0, 6.0)
        rgbs.append(f["rgb_map"])
        losses.append(O.view_loss(f["rgb_map"], b.ray_targets, chunk))
    return float(O.dataset_loss(losses)), [float(x) for x in losses], rgbs


def test_train_nerf_cli_trains_and_checkpoints(trained):
    """`train_nerf --config`: 12 optimizer steps through Trainer.fit on the HIP forward/backward, progress lines from
    LoggerCallback, one validation at step 8 and one at the end, Lightning's directory layout."""
    out, vdir = trained["stdout"], trained["vdir"]
    assert vdir.endswith(os.path.join("flow", "gpu", "version_0"))
    assert "[TRAIN] Iter: 4 LOSS:" in out and "[VAL] =======> Iter: 8" in out and "Done!" in out
    files = sorted(os.listdir(os.path.join(vdir, "checkpoints")))
    assert "model_last.ckpt" in files and any(f.startswith("model_epoch=") for f in files)
    flat = yaml.safe_load(open(os.path.join(vdir, "hparams.yaml")))
    assert flat["models.coarse.hidden_size"] == 64 and flat["experiment.train_iters"] == 12
    ck = torch.load(os.path.join(vdir, "checkpoints", "model_last.ckpt"), weights_only=False)
    assert ck["global_step"] == 12 and len(ck["state_dict"]) == 38 and len(ck["optimizer_states"]) == 1
    import json
    rows = [json.loads(l) for l in open(os.path.join(vdir, "metrics.jsonl"))]
    tl = [r["train/loss"] for r in rows if "train/loss" in r]
    assert len(tl) == 3 and all(np.isfinite(tl)) and tl[-1] < tl[0]
    assert trained["trainer"].max_steps == 12 and trained["trainer"].check_val_every_n_epoch == 2


def test_eval_nerf_cli_matches_the_oracle_on_the_checkpoint(trained, scene, tmp_path):
    """`eval_nerf --log-checkpoint ... --save-images --save-disparity`: BlenderDataset(TEST) read from the PNG files,
    rays generated on the GPU, 1600 rays in chunks of 700 through model.query; the dataset loss equals the oracle's
    (reference bookkeeping incl. the float batch count) on the checkpoint's weights, the PNGs are the renders."""
    from nerfmeshes_amd import eval_nerf
    from nerfmeshes_amd.data import BlenderDataset, DatasetType
    from nerfmeshes_amd.lightning_modules import PathParser
    _, stored = scene
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        total = eval_nerf.main(["--log-checkpoint", trained["vdir"], "--save-dir", str(tmp_path), "--save-images", "--save-disparity"])
        cfg, _ = PathParser().parse(None, trained["vdir"], None, "model_last.ckpt")
        ds = BlenderDataset(cfg, type=DatasetType.TEST)
    assert len(ds) == 3
    # the reader returns the stored 8-bit targets exactly
    assert torch.equal(ds[0]["ray_targets"], stored["test"][0])
    want, per_view, rgbs = _oracle_dataset_loss(os.path.join(trained["vdir"], "checkpoints", "model_last.ckpt"), trained["hp"], ds, 700)
    assert abs(float(total) - want) <= 1e-5 * want, (float(total), want)
    text = buf.getvalue()
    assert text.count("[EVAL] Iter:") == 3 and "Dataset loss MSE:" in text
    from PIL import Image
    img = np.asarray(Image.open(tmp_path / "flow" / "images" / "0001.png"))
    ref = (rgbs[1].view(SIZE, SIZE, 3).clamp(0, 1) * 255).to(torch.uint8).numpy()
    assert img.shape == (SIZE, SIZE, 3) and np.abs(img.astype(int) - ref.astype(int)).max() <= 1
    assert (np.abs(img.astype(int) - ref.astype(int)) > 0).mean() < 0.01          # only values that sit on a rounding edge
    tgt = np.asarray(Image.open(tmp_path / "flow" / "targets" / "0000.png"))
    assert np.array_equal(tgt, (stored["test"][0] * 255).to(torch.uint8).numpy())
    disp = np.asarray(Image.open(tmp_path / "flow" / "disparity" / "0000.png"))
    assert disp.shape == (SIZE, SIZE) and disp.dtype == np.uint8


def test_reference_shaped_eval_loop_call_by_call(trained):
    """The call sequence of /root/reference/src/eval_nerf.py:50-105 written out against the shim's names -- DataLoader
    batch -> DataBundle.deserialize(...).to_ray_batch() -> nerf.batchify(directions, targets, batch_size, device) ->
    model.query((origins.to(device), directions, HOST bounds)) -> F.mse_loss / float batch_count -> nerf.mse2psnr ->
    nerf.cast_to_disparity_image -- gives the mirror's dataset loss bit for bit, from a ray CACHE this time
    (`dataset.caching.use_caching`: written on first use with GPU-generated rays, then read back)."""
    import torch.nn.functional as F
    from torch.utils.data import DataLoader
    from nerfmeshes_amd import compat, eval_nerf
    models, nerf = compat.install()
    from data.datasets import BlenderDataset, DatasetType           # the names eval_nerf.py imports
    from data.data_helpers import DataBundle
    from lightning_modules import PathParser
    pp = PathParser()
    with contextlib.redirect_stdout(io.StringIO()):
        cfg, _ = pp.parse(None, trained["vdir"], None, "model_last.ckpt")
        cfg.dataset.caching.use_caching = True
        model = getattr(models, cfg.experiment.model).load_from_checkpoint(pp.checkpoint_path).eval().to("cuda")
        loader = DataLoader(BlenderDataset(cfg, type=DatasetType.TEST), batch_size=1)
    assert sorted(os.listdir(os.path.join(cfg.dataset.caching.cache_dir, "test"))) == ["0000.data", "0001.data", "0002.data"]
    losses = []
    with torch.no_grad():
        for ray_batch in loader:
            bundle = DataBundle.deserialize(ray_batch).to_ray_batch()
            batch_size = cfg.nerf.validation.chunksize
            batch_count = bundle.ray_directions.shape[0] / batch_size
            loss, disp = 0, []
            for dirs, tgts in nerf.batchify(bundle.ray_directions, bundle.ray_targets, batch_size=batch_size, device="cuda", progress=False):
                out = model.query((bundle.ray_origins.to("cuda"), dirs, bundle.ray_bounds))
                assert not bundle.ray_bounds.is_cuda
                loss += F.mse_loss(out.rgb_map, tgts)
                disp.append(out.disp_map)
            losses.append(loss / batch_count)
            image = nerf.cast_to_disparity_image(torch.cat(disp).view(bundle.hwf[0], bundle.hwf[1]), white_background=True)
            assert image.shape == (SIZE, SIZE)
        total = torch.stack(losses).mean()
        psnr = nerf.mse2psnr(total)
        class A:
            save_dir, save_images, save_disparity, synthesis_images = ".", False, False, False
        with contextlib.redirect_stdout(io.StringIO()):
            mirror = eval_nerf.eval_nerf(model, A, cfg, "cuda")
    assert torch.equal(total, mirror) and torch.isfinite(psnr)


def test_eval_nerf_synthesis_images(trained, tmp_path):
    """`--synthesis-images` (eval_nerf.py:29-31, datasets.py:103-133): 120 novel views on the 3-degree orbit, no losses."""
    from nerfmeshes_amd import eval_nerf
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        total = eval_nerf.main(["--log-checkpoint", trained["vdir"], "--save-dir", str(tmp_path), "--save-images", "--synthesis-images"])
    assert total is None and "Synthesizing dataset" in buf.getvalue() and "[EVAL]" not in buf.getvalue()
    files = sorted(os.listdir(tmp_path / "flow" / "images"))
    assert len(files) == 120 and files[0] == "0000.png" and not os.listdir(tmp_path / "flow" / "targets")


def test_train_nerf_resume_from_log_checkpoint(trained):
    """`train_nerf --log-checkpoint <version dir>` restores weights, optimizer and step counter and trains on."""
    from nerfmeshes_amd import train_nerf
    vdir = trained["vdir"]
    hp_path = os.path.join(vdir, "hparams.yaml")
    flat = yaml.safe_load(open(hp_path))
    flat["experiment.train_iters"] = 16
    open(hp_path, "w").write(yaml.safe_dump(flat))
    before = torch.load(os.path.join(vdir, "checkpoints", "model_last.ckpt"), weights_only=False)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        trainer, model, _ = train_nerf.main(["--log-checkpoint", vdir])
    after = torch.load(os.path.join(vdir, "checkpoints", "model_last.ckpt"), weights_only=False)
    assert after["global_step"] == before["global_step"] + 4 == 16 and trainer.global_step == 16
    assert any(not torch.equal(before["state_dict"][k], after["state_dict"][k]) for k in before["state_dict"])
    assert after["optimizer_states"][0]["state"][0]["step"] >= 15


@pytest.mark.parametrize("deterministic", [False, True])
def test_train_nerf_two_ranks_keep_identical_replicas(scene, deterministic):
    """`train_nerf` under torch.distributed.run.  Two ranks on the one GPU (NERFMESHES_RANKS_PER_GPU=2, gloo)
    run the reference's command line; without --deterministic every rank builds its own randomly initialised model, with it
    every rank seeds alike: either way the ranks must end with EQUAL parameters (rank 0's broadcast + averaged gradients),
    must have trained on DIFFERENT rays, and must share one version directory; rank 0's checkpoint is that model."""
    import subprocess
    import sys
    import socket
    root, _ = scene
    cfg_path, hp = _config(root, **{"experiment.id": f"ddp{int(deterministic)}", "experiment.train_iters": 8, "experiment.validate_every": 8})
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(NERFMESHES_RANKS_PER_GPU="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(here, "tools", "train_worker.py"), cfg_path, str(int(deterministic))],
                       cwd=os.path.dirname(here), env=env, capture_output=True, text=True, timeout=900)
    tb = r.stderr[r.stderr.find("Traceback"):][:3000] if "Traceback" in r.stderr else r.stderr[-3000:]
    assert r.returncode == 0, r.stdout[-1500:] + tb
    assert "TRAIN_DDP_OK world=2 steps=8" in r.stdout, r.stdout[-1500:]
