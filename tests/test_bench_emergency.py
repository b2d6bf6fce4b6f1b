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
