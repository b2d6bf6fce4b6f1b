"""Route A of INTEGRATION.md, executed: the UNMODIFIED reference scripts run as `__main__` over `nerfmeshes_amd.compat.install()`.

Each scenario runs in a fresh interpreter (tests/tools/ref_script_runner.py) because the shim registers itself under the
reference's module names.  No GPU here, so the model class the checkpoints name is a test double whose arithmetic is the
CPU oracle; what these tests pin is the whole surface between the scripts and the kernels.  Skipped
where /root/reference does not exist (the GPU box): tests/test_gpu_reference_flow.py covers the mirrors there."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("NERFMESHES_REFERENCE", "/root/reference")

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src")), reason="reference tree not present")


def _run(scenario, tmp_path):
    env = dict(os.environ, CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", "ref_script_runner.py"), scenario, str(tmp_path)],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_reference_scripts_import_under_the_shim(tmp_path):
    """`from nerf import export_point_cloud`, `from data.datasets import BlenderDataset, DatasetType`,
    `pytorch_lightning.core.memory` ... every import of the three scripts resolves; third-party packages missing offline
    come from the stand-ins compat.install() registers."""
    out = _run("imports", tmp_path)
    assert "eval_nerf" in out["eval_nerf"] and "export_point_cloud" in out["eval_nerf"] and "BlenderDataset" in out["eval_nerf"]
    assert {"extract_radiance", "extract_iso_level", "extract_geometry", "export_marching_cubes"} <= set(out["mesh_nerf"])
    assert {"main", "Trainer", "LoggerCallback", "ModelCheckpoint"} <= set(out["train_nerf"])
    assert "pytorch_lightning" in out["stand_ins"] and "skimage.measure" in out["stand_ins"]


def test_reference_eval_nerf_main(tmp_path):
    """eval_nerf.py as __main__: PathParser -> load_from_checkpoint -> BlenderDataset(TEST) over a ray cache ->
    DataLoader -> batchify -> model.query -> loss with the float batch count -> PNGs.  The numbers it prints are the
    oracle bookkeeping's, and the package's mirror `nerfmeshes_amd.eval_nerf.eval_nerf` (same signature) reproduces the
    script's loss exactly and its image files byte for byte."""
    out = _run("eval", tmp_path)
    lines = out["stdout"].splitlines()
    for i, loss in enumerate(out["expected_losses"]):
        assert any(l.startswith(f"[EVAL] Iter: {i} Loss MSE {loss} ") for l in lines), (loss, lines)
    assert any(l.startswith(f"Dataset loss MSE: {out['expected_total']} ") for l in lines), lines
    assert out["mirror_total"] == out["expected_total"]
    assert [l for l in out["mirror_stdout"].splitlines() if "EVAL" in l or "Dataset loss" in l] == \
           [l for l in lines if "EVAL" in l or "Dataset loss" in l]
    assert len(out["files"]) == 6 and out["mirror_files_identical"] and out["image0_matches_render"]


def test_reference_mesh_nerf_main(tmp_path):
    """mesh_nerf.py as __main__ (both appearance branches): extract_radiance's batchify loop over `model.sample_points`,
    numpy iso level, `skimage.measure.marching_cubes` (here the C oracle behind the same name), per-vertex re-query
    through `model.query` with per-ray origins and host bounds, mesh cache, `nerf.export_obj` (the native writer)."""
    out = _run("mesh", tmp_path)
    for tag in ("view", "diffuse"):
        assert out[tag]["v"] == out[tag]["vn"] == out["expected"]["v"] > 100
        assert out[tag]["f"] == out["expected"]["f"] > 100
        assert out[tag]["cache"] and "Finished writing" in out[tag]["stdout"]
        assert f"Querying based on iso level: {out['expected']['iso']}" in out[tag]["stdout"]
    assert out["view"]["first_v"].split()[:4] == out["diffuse"]["first_v"].split()[:4]      # same geometry ...
    assert out["view"]["first_v"] != out["diffuse"]["first_v"]                                # ... different colours


def test_reference_train_nerf_main(tmp_path):
    """train_nerf.py as __main__ from a nested yml, then resumed with --log-checkpoint: logger + version dir,
    BaseModel.setup -> datasets / trainer sizing, train / val dataloaders, training_step -> backward -> Adam -> per-step
    scheduler, LoggerCallback lines, validation every `validate_every` steps, ModelCheckpoint file names, a checkpoint
    that `load_from_checkpoint` reads back."""
    out = _run("train", tmp_path)
    assert "[TRAIN] Iter: 2 LOSS:" in out["stdout"] and "[VAL] =======> Iter: 3" in out["stdout"] and "Done!" in out["stdout"]
    assert "model_last.ckpt" in out["checkpoints"] and "model_epoch=0.ckpt" in out["checkpoints"]
    assert {"state_dict", "hyper_parameters", "optimizer_states", "lr_schedulers", "epoch", "global_step"} <= set(out["checkpoint_keys"])
    assert out["hparams_yaml"] and out["state_dict_keys"] == 38 and out["reloaded_params"] > 0
    assert out["train_losses"][-1] < out["train_losses"][0]
    # resumed with train_iters 10 on 3 training images: BaseModel.setup sizes the run as 10 // 3 = 3 epochs = 9 steps
    assert out["global_step"] == 6 and out["resumed_global_step"] == 9 and out["weights_moved"]
    assert "[TRAIN] Iter: 8" in out["resume_stdout"]


@pytest.mark.parametrize("scenario", ["eval", "mesh", "train"])
def test_the_committed_call_traces_are_what_the_unmodified_scripts_do(scenario, tmp_path):
    """tests/golden/script_traces.json (the fixture tests/test_gpu_script_traces.py holds the package's command lines to on the
    MI355X) regenerates from the UNMODIFIED reference scripts: the tiny shapes are re-run here (seconds) and must reproduce the
    committed trace and printed numbers exactly; the shipped-shape entries (minutes of CPU: tests/golden/make_script_traces.py) are
    checked for the structure the scripts' own loops imply."""
    env = dict(os.environ, CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="", NM_REF_BACKEND="oracle", NM_REF_WHICH="reference", NM_REF_SHAPES="tiny")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", "script_trace_runner.py"), scenario, str(tmp_path)],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    got = json.loads(r.stdout.strip().splitlines()[-1])
    fixture = json.load(open(os.path.join(ROOT, "tests", "golden", "script_traces.json")))
    want = fixture["tiny"][scenario]
    assert got["trace"] == want["trace"]
    # (numbers to 1e-5: torch's CPU matrix kernels block by thread count, which a different host may set differently)
    for key in ("stdout_losses", "stdout_total", "train_losses", "eval_losses"):
        if key in want:
            assert len(got[key]) == len(want[key]) and all(abs(a - b) <= 1e-5 * max(1.0, abs(b)) for a, b in zip(got[key], want[key])), key
    for key in ("global_step", "resumed_global_step", "checkpoints"):
        if key in want:
            assert got[key] == want[key], key
    if scenario == "mesh":
        assert {k: got["view"][k] for k in ("v", "f")} == {k: want["view"][k] for k in ("v", "f")}
        assert abs(got["view"]["iso"][0] - want["view"]["iso"][0]) <= 1e-5 * abs(want["view"]["iso"][0])
    calls = lambda trace, name: sum(c for (n, _, _), c in trace if n == name)  # noqa: E731
    ship = fixture["shipped"][scenario]["trace"]
    if scenario == "eval":        # two 100 x 100 views in chunks of 2048 rays: 5 queries each (eval_nerf.py:62-65)
        assert calls(ship, "query") == 10 and calls(ship, "load_from_checkpoint") == 1
    elif scenario == "mesh":      # 128^3 points at the script's own --batch-size 1024, twice (mesh_nerf.py:37-48,239)
        assert calls(ship, "sample_points") >= 2 * 2048 and calls(ship, "marching_cubes") == 2 and calls(ship, "export_obj") == 2
    else:                          # 18 + 6 steps, a validation pass per epoch of 3 views, then eval_nerf.py on the result
        assert calls(ship, "training_step") == 24 and calls(ship, "configure_optimizers") == 2 and calls(ship, "query") > 0
