"""Writes tests/golden/script_traces.json: the call traces and printed numbers of the UNMODIFIED reference scripts
(/root/reference/src/{eval_nerf,mesh_nerf,train_nerf}.py run as __main__ over nerfmeshes_amd.compat.install(), arithmetic = the
CPU oracle) at the shipped shapes and at the tiny ones -- the fixture tests/test_gpu_script_traces.py holds the package's command
lines to on the MI355X.  Container only (needs /root/reference); about ten minutes of CPU.

    python tests/golden/make_script_traces.py [--only eval,mesh,train] [--shapes shipped,tiny]
"""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(ROOT, "tests", "golden", "script_traces.json")


def run(scenario, shapes):
    env = dict(os.environ, CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="", NM_REF_BACKEND="oracle", NM_REF_WHICH="reference",
               NM_REF_SHAPES=shapes)
    with tempfile.TemporaryDirectory() as work:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", "script_trace_runner.py"), scenario, work],
                           cwd=ROOT, env=env, capture_output=True, text=True)
    if r.returncode != 0:
        raise SystemExit(r.stdout[-2000:] + r.stderr[-4000:])
    return json.loads(r.stdout.strip().splitlines()[-1])


if __name__ == "__main__":
    only = sys.argv[sys.argv.index("--only") + 1].split(",") if "--only" in sys.argv else ["eval", "mesh", "train"]
    shapes = sys.argv[sys.argv.index("--shapes") + 1].split(",") if "--shapes" in sys.argv else ["shipped", "tiny"]
    data = json.load(open(OUT)) if os.path.exists(OUT) else {}
    data["what"] = ("call traces (run-length encoded: [[name, args, kwargs], count]) and printed numbers of the UNMODIFIED reference "
                    "scripts over compat.install() with the CPU oracle as arithmetic; written by tests/golden/make_script_traces.py "
                    "from /root/reference/src (never copied); tests/tools/script_trace_runner.py documents the entries")
    for sh in shapes:
        for sc in only:
            res = run(sc, sh)
            for k in ("repacks", "backend", "which", "shapes", "marching_cubes_is_stand_in"):
                res.pop(k, None)
            data.setdefault(sh, {})[sc] = res
            print(sh, sc, "trace entries:", len(res["trace"]), "calls:", sum(c for _, c in res["trace"]), flush=True)
            with open(OUT, "w") as fh:
                json.dump(data, fh, indent=1)
