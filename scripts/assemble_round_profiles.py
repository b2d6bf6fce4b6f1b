"""After `bash scripts/profile_round_lean.sh <tag>` (through gpurun) and `python scripts/summarize_profiles.py <tag>`: copy what the
round's claims rest on from gpurun_out/ (scratch) into profiles/ (committed).

    python scripts/assemble_round_profiles.py r06
"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")


def rows(path):
    out = {}
    if os.path.exists(path):
        for line in open(path):
            name, brace, rest = line.partition(" {")
            if brace and not line.startswith("/"):
                try:
                    out[name] = json.loads("{" + rest)
                except ValueError:
                    pass
    return out


shutil.copy(os.path.join(G, f"bench_{tag}.json"), os.path.join(P, f"{tag}_bench_line.json"))           # the line as printed
shutil.copy(os.path.join(G, f"bench_full_{tag}.json"), os.path.join(P, f"{tag}_bench_full.json"))      # every object in full
full = json.load(open(os.path.join(P, f"{tag}_bench_full.json")))

before = rows(os.path.join(P, f"{tag}_raw", "train_shapes_before.txt"))
json.dump({
    "what": "whole training iterations of the shipped networks besides the headline's (tests/tools/bench_train_shapes.py: ms per iteration "
            "eager, whole-iteration fraction of the fp32 MFMA peak, per-stage HIP-event medians; final tree of the round, one box) and the same "
            "shapes from the bench line's train.shapes / tiny.train objects (eager + replayed from one hipGraph, fused Adam)",
    "peak_tflops": 157.3,
    "shapes": rows(os.path.join(G, f"train_shapes_{tag}.txt")),
    "start_of_the_round_same_tool": {k: {kk: v[kk] for kk in ("ms_per_iteration", "frac_of_fp32_mfma_peak_whole_iteration", "stage_ms")}
                                      for k, v in before.items()},
    "bench_line_objects": {"train.shapes": full.get("train", {}).get("shapes"), "tiny.train": full.get("tiny", {}).get("train")},
    "what_changed": "one partial per workgroup (k-split waves add up in LDS), a tuned 64x64 weight-gradient kernel, 32-row chunks for the "
                    "<= 128-wide products, the encoding rows written by the taping forward (no separate encode pass: the 'encodings' stage is "
                    "gone) -- DESIGN.md section 3.5, round-6 paragraph",
    "ceilings": "profiles/r06_pmc_dw_kernels.json (the 128-wide weight gradients are clock-bound: matrix pipe 84 % busy at 1.80 GHz), "
                "profiles/r06_raw/dma_ring_probe.txt (what the dataflow reaches on its own)",
}, open(os.path.join(P, f"{tag}_generic_train.json"), "w"), indent=1)

json.dump({
    "what": "the package's command lines (python -m nerfmeshes_amd.{eval,mesh,train}_nerf; mesh_nerf --route script) over the HIP kernels on "
            "1 x MI355X, held to the call traces and printed numbers of the UNMODIFIED reference scripts (/root/reference/src/{eval_nerf,"
            "mesh_nerf,train_nerf}.py run as __main__ in the build container over compat.install(), arithmetic = the CPU oracle: "
            "tests/golden/script_traces.json); tests/test_gpu_script_traces.py, 6 passed on the final tree",
    "why_not_the_scripts_themselves": "the reference checkout may not travel to the GPU box in any form and the build container has no GPU; "
                                      "round 4's one-off staging of the three script files next to the tree is not repeated.  What runs on the "
                                      "GPU instead of the scripts' ~100 top-level lines are the package's command lines, and the trace (every call "
                                      "across the model API with the shapes / dtypes of its arguments) proves they ask the same shim classes and "
                                      "kernels for the same work",
    "shipped_shapes": json.load(open(os.path.join(G, "script_traces_shipped.json"))),
    "tiny_shapes": json.load(open(os.path.join(G, "script_traces_tiny.json"))),
}, open(os.path.join(P, f"{tag}_reference_scripts_on_hip.json"), "w"), indent=1)
print("headline", full["value"], "rays/s, frac", full["roofline"]["frac"], "| train", full["train"]["ms_per_iteration"], "ms",
      "| 8x128", full["train"]["shapes"]["8x128"]["ms_per_iteration"], "| tiny.train replay", full["tiny"]["train"]["ms_per_iteration_graph_replay"])
