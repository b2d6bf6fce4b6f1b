"""GPU: the package's three command lines, over the real kernels at the shapes the reference ships, held to what the UNMODIFIED
reference scripts did -- call for call and number for number.

`tests/golden/script_traces.json` holds, for `eval_nerf.py`, `mesh_nerf.py` and `train_nerf.py` run as `__main__` from
/root/reference/src in the build container (over `compat.install()`, arithmetic = the CPU oracle): every call that crossed the
model API with the shapes of its arguments (the trace), and the numbers the script printed.  The reference cannot travel to the
GPU box, so the scripts' own top-level lines are the one thing that cannot execute next to the kernels; here the package's
command lines (same flags; `mesh_nerf --route script` = the script's own call sequence) run in their place on the MI355X and must
produce the IDENTICAL trace -- same calls, same order, same shapes, same dtypes -- and the script's printed losses / iso levels /
vertex counts within the render tolerance (tests/tools/script_trace_runner.py explains the set-up; the 4x32 `tiny` shapes are the
ones tests/test_reference_scripts.py runs on the CPU).

A report of every run (wall time per script, ABI version, how often the parameters were re-packed, the numbers side by side) goes
to gpurun_out/script_traces_<shapes>.json; profiles/r06_reference_scripts_on_hip.json is a committed copy."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIXTURE = json.load(open(os.path.join(ROOT, "tests", "golden", "script_traces.json")))

pytestmark = pytest.mark.gpu
LOSS_TOL = 1e-5          # relative, on a per-view MSE / the dataset MSE: the render tolerance (2e-5 abs per pixel value) squared away


def _run(scenario, shapes, tmp_path):
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a MI355X")
    env = dict(os.environ, NM_REF_BACKEND="hip", NM_REF_WHICH="mirror", NM_REF_SHAPES=shapes)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", "script_trace_runner.py"), scenario, str(tmp_path)],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["backend"] == "hip" and out["which"] == "mirror" and out["native_library"] == "libnerfmeshes_hip.so"
    _report(shapes, scenario, out)
    return out, FIXTURE[shapes][scenario]


def _report(shapes, scenario, out):
    path = os.path.join(ROOT, "gpurun_out", f"script_traces_{shapes}.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    data = json.load(open(path)) if os.path.exists(path) else {}
    want = FIXTURE[shapes][scenario]
    entry = {k: v for k, v in out.items() if k != "trace"}
    entry["trace_calls"] = sum(c for _, c in out["trace"])
    entry["trace_identical_to_the_unmodified_script"] = out["trace"] == want["trace"]
    if scenario == "mesh":        # the re-query's call count / ragged tail follow the vertex count (see _counts_out)
        batch = {"shipped": 1024, "tiny": 3000}[shapes]
        entry["trace_identical_up_to_vertex_count_dependent_batches"] = _counts_out(out["trace"], batch) == _counts_out(want["trace"], batch)
    entry["trace_names"] = [[name, count] for (name, _, _), count in out["trace"]]
    entry["reference_script_printed"] = {k: v for k, v in want.items() if k not in ("trace", "wall_s", "files")}
    data[scenario] = entry
    with open(path, "w") as fh:
        json.dump(data, fh, indent=1)


def _same_trace(got, want):
    assert len(got) == len(want), (len(got), len(want), [e for e in got[:6]], [e for e in want[:6]])
    for i, (a, b) in enumerate(zip(got, want)):
        assert a == b, (i, a, b)


def _counts_out(trace, batch=1024):
    """mesh_nerf's re-query walks the V vertices in --batch-size calls: how many full calls there are and how long the ragged last
    one is depends on V, which may differ by the end-to-end budget (a sign flip of one voxel within fp32 round-off of the level).
    Everything up to and including marching_cubes is compared as it is; behind it, call counts of full batches may differ by one
    and a ragged batch is compared as "ragged"."""
    out, tail = [], False
    for (name, args, kwargs), count in trace:
        if tail and name in ("query", "sample_points"):
            lead = args[0][0][1][0] if name == "query" else args[0][1][0]
            if lead != batch:
                args = "ragged"
            else:
                count = "full batches"
        out.append([[name, args, kwargs], count])
        tail = tail or name == "marching_cubes"
        if name == "load_from_checkpoint":
            tail = False
    return out


def _close(a, b, tol=LOSS_TOL):
    assert len(a) == len(b), (a, b)
    for x, y in zip(a, b):
        assert abs(x - y) <= tol * max(1.0, abs(y)), (x, y)


@pytest.mark.parametrize("shapes", ["shipped", "tiny"])
def test_eval_nerf_asks_the_kernels_what_the_reference_script_asks(shapes, tmp_path):
    """eval_nerf.py (/root/reference/src/eval_nerf.py:50-105): load_from_checkpoint, then `model.query` once per chunk of
    cfg.nerf.validation.chunksize rays of every test view (shipped: 8x256, 64 + 128 samples, two 100 x 100 views in chunks of 2048
    = 5 calls per view, the last one ragged).  Same trace; the per-view and dataset losses the reference's script printed over the
    oracle equal the ones printed here over nm_render_rays to 1e-5; same image files."""
    out, want = _run("eval", shapes, tmp_path)
    _same_trace(out["trace"], want["trace"])
    _close(out["stdout_losses"], want["stdout_losses"])
    _close(out["stdout_total"], want["stdout_total"])
    assert out["files"] == want["files"] and len(out["stdout_losses"]) == 2
    # one handle per network, built once; under the default guard every query re-packs both (a 5 us gather each)
    calls = sum(c for (name, _, _), c in out["trace"] if name == "query")
    assert out["repacks"] == [[calls, calls]]


@pytest.mark.parametrize("shapes", ["shipped", "tiny"])
def test_mesh_nerf_asks_the_kernels_what_the_reference_script_asks(shapes, tmp_path):
    """mesh_nerf.py (/root/reference/src/mesh_nerf.py:27-53,68-92,131-201) at its OWN defaults (shipped: --res 128 --iso-level 32
    --limit 1.2, --batch-size 1024: 2048 sample_points calls for the grid, one skimage.measure.marching_cubes, ~50 re-query calls,
    export_obj), both appearance branches.  Same trace.  The iso level is the script's to 1e-4; vertex and face counts are EQUAL
    when no voxel of the grid lies within fp32 round-off of the level, and within the end-to-end budget of DESIGN.md (a sign flip
    moves at most 8 cut cubes) otherwise -- nm_mc_* itself is bitwise on an identical grid (tests/test_gpu_mc.py)."""
    out, want = _run("mesh", shapes, tmp_path)
    batch = {"shipped": 1024, "tiny": 3000}[shapes]                 # the script's own default / the tiny scenario's --batch-size
    _same_trace(_counts_out(out["trace"], batch), _counts_out(want["trace"], batch))
    for tag in ("view", "diffuse"):
        g, w = out[tag], want[tag]
        assert g["finished"] and g["cache"] and g["v"] == g["vn"]
        assert abs(g["iso"][0] - w["iso"][0]) <= 1e-4 * max(1.0, abs(w["iso"][0])), (g["iso"], w["iso"])
        assert abs(g["v"] - w["v"]) <= max(2, w["v"] // 5000) and abs(g["f"] - w["f"]) <= max(4, w["f"] // 5000), (g["v"], w["v"], g["f"], w["f"])
        assert g["first_f"] == w["first_f"]
    if shapes == "tiny":
        assert (out["view"]["v"], out["view"]["f"]) == (want["view"]["v"], want["view"]["f"])


@pytest.mark.parametrize("shapes", ["shipped", "tiny"])
def test_train_nerf_asks_the_kernels_what_the_reference_script_asks(shapes, tmp_path):
    """train_nerf.py (/root/reference/src/train_nerf.py:62-101) from a nested yml, resumed from the log directory, then eval_nerf.py on
    the resumed checkpoint (shipped: nerf-colmap-fern's 8x128, 64 + 128 samples, 1024 random rays, 20 + 4 steps of the FUSED Adam
    make_optimizer picks on the GPU).  Same trace (configure_optimizers, every training_step / validation_step with the shapes of
    its batch, the resumed run, the evaluation's queries).  The random ray selection draws from the device generator here and
    from the host generator in the container, so the training losses are not comparable number for number: they must stay
    finite and bounded (the reference's own run over the oracle does not fall monotonically in 18 steps either), the resumed run
    must move the weights, and the evaluation of the trained checkpoint must equal the ORACLE's evaluation of
    that same checkpoint (computed on this box) to 1e-5 -- the test that a model which trained on stale weights would fail."""
    out, want = _run("train", shapes, tmp_path)
    _same_trace(out["trace"], want["trace"])
    assert out["done"] and out["checkpoints"] == want["checkpoints"] and out["state_dict_keys"] == want["state_dict_keys"]
    assert (out["global_step"], out["resumed_global_step"]) == (want["global_step"], want["resumed_global_step"])
    assert out["weights_moved"] and all(0.0 < x < 1.05 * out["train_losses"][0] for x in out["train_losses"]), out["train_losses"]
    assert [l.split(" LOSS")[0] for l in out["train_lines"]] == [l.split(" LOSS")[0] for l in want["train_lines"]]
    _close(out["eval_losses"], out["oracle_eval_losses_on_this_checkpoint"])
