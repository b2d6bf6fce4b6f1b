"""CPU: bench.py's emergency writer (the ONE JSON line must reach stdout even when the launcher tears the job down around
rank 0 while its main thread sits in a blocking call) and the store-based verdict on a secondary object -- without a GPU.
The GPU side (two ranks sharing the device, a failure injected into `mesh` / `buff`) is tests/test_gpu_dist.py."""
import json
import os
import signal
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys, time, threading
sys.path.insert(0, %r)
import bench
fd = os.dup(1)
em = bench._Emergency(fd)
em.arm({"metric": "m", "value": 1.0, "mesh": {"ok": True}})
em.stage = "buff"
sys.stderr.write("armed\n"); sys.stderr.flush()
# the main thread blocks in C with the GIL released, as a collective / device synchronisation does: a Python-level signal
# handler could not run here
threading.Event().wait(60)
sys.stderr.write("not reached\n")
'''


def test_emergency_writer_emits_the_line_on_sigterm_while_the_main_thread_is_blocked():
    p = subprocess.Popen([sys.executable, "-c", CHILD % ROOT], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT)
    assert p.stderr.readline().strip() == "armed"
    time.sleep(0.2)
    p.send_signal(signal.SIGTERM)
    out, err = p.communicate(timeout=30)
    assert p.returncode == 3 and "not reached" not in err
    lines = [ln for ln in out.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    line = json.loads(lines[0])
    assert line["value"] == 1.0 and line["mesh"] == {"ok": True}, "what was measured before the failure is on the line"
    assert "terminated during 'buff'" in line["buff"]["error"] and line["errors"] == [line["buff"]["error"]]


def test_guarded_object_reports_its_error_and_keeps_the_rest():
    sys.path.insert(0, ROOT)
    import bench
    em = bench._Emergency(os.dup(1))
    ok = bench._guarded("tiny", lambda: {"value": 2.0}, 0, 1, em)
    bad = bench._guarded("train", lambda: 1 / 0, 0, 1, em)
    assert ok == {"value": 2.0} and "ZeroDivisionError" in bad["error"] and em.stage == "train"
    os.environ["NM_BENCH_INJECT_FAILURE"] = "eval:0"
    try:
        assert "injected failure" in bench._guarded("eval", lambda: {"value": 3.0}, 0, 1, em)["error"]
    finally:
        del os.environ["NM_BENCH_INJECT_FAILURE"]
    out = {"cpu_baseline": {"kind": "port", "value": 1.0}, "nested": [{"kind": "port"}, {"kind": "reference"}]}
    bench._annotate_ports(out)
    assert out["cpu_baseline"]["port_over_reference_time"] == bench.PORT_OVER_REFERENCE_TIME
    assert "port_over_reference_time" in out["nested"][0] and "port_over_reference_time" not in out["nested"][1]


def test_the_committed_bench_line_keeps_the_drivers_contract():
    """profiles/r06_bench_line.json is the line bench.py printed on the MI355X (the compact form), profiles/r06_bench_full.json the
    objects of the same run in full: the keys the driver and the judge read must all be on the LINE, typed as the contract says, and
    the derived figures must follow from the primary ones (frac = achieved / peak; value = rays per step / time per step; the CPU
    leg says what it timed); the line fits the driver's record."""
    text = open(os.path.join(ROOT, "profiles", "r06_bench_line.json")).read()
    assert len(text.strip()) <= 4096 and len(text.strip().splitlines()) == 1
    line = json.loads(text)
    full = json.load(open(os.path.join(ROOT, "profiles", "r06_bench_full.json")))
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    for key, kind in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int), ("ms_per_step", float),
                      ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str), ("config", dict), ("roofline", dict),
                      ("cpu_baseline", dict)):
        assert isinstance(line[key], kind), (key, type(line[key]))
        assert full[key] == line[key] or isinstance(line[key], dict), key
    assert "vs_baseline" in line and line["higher_is_better"] is True and line["scaling"] == "weak" and line["n_gpus"] == 1
    assert "workload" in line["config"] and "model" not in line["config"]
    assert line["unit"] == base.get("unit", line["unit"]) and line["full"] == "bench_full.json"
    assert abs(line["value"] - line["config"]["rays_per_step_per_rank"] / line["ms_per_step"] * 1e3) < 1e-6 * line["value"]
    roof = line["roofline"]
    assert roof["bound"] in ("hbm", "mfma") and roof["unit"] in ("GB/s", "TFLOP/s")
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-4 and 0.0 < roof["frac"] <= 1.0
    assert roof["traffic"] is None or roof["traffic"] > 0
    cpu = line["cpu_baseline"]
    assert cpu["kind"] in ("reference", "port") and cpu["value"] > 0 and cpu["cores"] >= 1 and isinstance(cpu["sample"], str) and cpu["sample"]
    assert cpu["kind"] != "port" or "port_over_reference_time" in cpu
    assert line["parity"]["abs_dpsnr_db"] <= 1e-4 and line["parity"]["unexplained"] == 0
    for name in ("train", "tiny", "mesh", "buff", "eval"):       # the secondary objects of the line, none of them an error
        assert isinstance(line[name], dict) and "error" not in line[name], name
        assert isinstance(full[name], dict) and "error" not in full[name], name
    assert not line.get("errors")
    assert line["mesh"]["marching_cubes"]["bitwise"] is True and line["mesh"]["grid_query"]["at_reference_batch_1024"]["same_sigma_as_one_call"] is True
    assert set(line["train"]["shapes"]) == {"8x128", "8x64"} and 0.5 < line["train"]["shapes"]["8x128"]["frac"] < 1.0
    train = full["train"]
    assert abs(train["roofline"]["frac"] - train["roofline"]["achieved"] / train["roofline"]["peak"]) < 1e-9
    assert abs(sum(k["ms"] for k in train["kernels"].values()) - train["ms_per_iteration"]) < 0.6, "the breakdown adds up to the iteration"
    assert "encodings" not in train["kernels"], "the taping forward writes the encoding rows: no separate pass (round 6)"


def test_the_printed_line_is_compact_and_keeps_the_contract():
    """What bench.py prints since round 6 (benchlib/line.py): the contract's keys untouched, `roofline` / `cpu_baseline` / `parity`,
    one small object per secondary figure -- at most 4 KiB, so that the driver's record holds it whole (round 5's 15 KB line
    survived there as key names and two truncated tails) -- and the name of the file that holds every object in full.  Fed with
    a full set of objects as measured on the MI355X (profiles/r05_bench_line.json)."""
    sys.path.insert(0, ROOT)
    from benchlib import line as L
    full = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_line.json")))
    text = L.render(full, "bench_full.json")
    assert len(text) <= L.LINE_BUDGET_BYTES and "\n" not in text
    line = json.loads(text)
    for key in L.CONTRACT:
        assert line[key] == full[key], key
    assert line["full"] == "bench_full.json"
    roof = line["roofline"]
    assert roof["bound"] == "mfma" and roof["unit"] == "TFLOP/s" and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-4
    assert roof["traffic"] > 0 and roof["avg_launch_ms"] > 0 and roof["kernel"].startswith("nm::")
    cpu = line["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["cores"] >= 1 and cpu["value"] > 0 and cpu["sample"] and cpu["port_over_reference_time"] == 1.15
    assert line["parity"]["abs_dpsnr_db"] <= 1e-4 and line["parity"]["unexplained"] == 0
    # the figures a reviewer needs from each secondary object, without opening the full file
    assert line["mesh"]["marching_cubes"]["bitwise"] is True and line["mesh"]["marching_cubes"]["bound"] == "hbm"
    assert 0 < line["mesh"]["grid_query"]["frac"] <= 1 and line["mesh"]["topology_128"]["ok"] is True
    for name in ("buff", "eval", "tiny", "train", "bf16x3"):
        assert line[name]["value"] > 0 and "unit" in line[name], name
    for name in ("buff", "eval", "tiny", "train"):
        assert 0 < line[name]["frac"] <= 1, name
    assert set(line["train"]["stages"]) >= {"taping_forward", "delta", "weight_gradients"}
    # an object that failed stays an error; one that outgrows the budget is cut to its scalars, never dropped
    full["buff"] = {"error": "RuntimeError('x')", "failed_ranks": [1]}
    full["train"]["stages_blowup"] = {str(i): {"ms": float(i)} for i in range(400)}
    full["train"]["kernels"].update({f"stage{i}": {"ms": 1.0 + i, "frac": 0.5} for i in range(200)})
    text = L.render(full, "bench_full.json")
    line = json.loads(text)
    assert len(text) <= L.LINE_BUDGET_BYTES and line["buff"] == {"error": "RuntimeError('x')", "failed_ranks": [1]}
    assert line["train"]["value"] == 145040.0 and "stages" not in line["train"]
