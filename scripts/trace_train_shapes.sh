#!/bin/bash
# rocprofv3 kernel trace of a few whole training iterations per network shape (run on the GPU box through gpurun): every
# kernel an iteration launches, by name; a library GEMM / reduction (rocBLAS Cijk_*, at::native reduce / gemm kernels)
# between the forward and the optimizer is a failure.  Summary -> gpurun_out/train_trace/summary.json
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/train_trace
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for shape in "4x64" "4x64 (config 1: 32 coarse, no fine)" "8x64" "8x100" "8x320" "4x400 flat" "8x256 ragged (583 rays)" "8x256"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --stats -d $out/raw$i -o t --output-format csv -- python $GRAFT_REPO_ROOT/tests/tools/bench_train_shapes.py --trace-one "$shape" > $out/log$i.txt 2>&1
    echo "$shape" > $out/raw$i/shape.txt
done
python - <<PY
import csv, glob, json, os, re
out = "$out"
res = {}
banned = re.compile(r"Cijk_|rocblas|gemm|at::native::reduce_kernel|at::native::.*[Ss]um", re.I)
for d in sorted(glob.glob(out + "/raw*")):
    shape = open(d + "/shape.txt").read().strip()
    names = {}
    for f in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            names[r["Name"]] = {"calls": int(r["Calls"]), "total_us": round(float(r["TotalDurationNs"]) / 1e3, 1)}
    # the one reduction torch itself runs in an iteration: the mean inside torch.nn.functional.mse_loss over the (R, 3) pixels --
    # the caller's LOSS (reference model_nerf.py:112-118), not a gradient; listed on its own
    loss = sorted(n for n in names if banned.search(n) and "MeanOps" in n)
    bad = sorted(n for n in names if banned.search(n) and "MeanOps" not in n)
    top = sorted(names.items(), key=lambda kv: -kv[1]["total_us"])
    res[shape] = {"kernels": len(names), "library_gemm_or_reduce_kernels": bad, "mse_loss_mean_kernels_of_the_caller": loss,
                  "by_time": [{"name": re.sub(r"\(.*", "", n)[:90], **v} for n, v in top]}
    print(shape, "kernels:", len(names), "library GEMM / gradient-reduction kernels:", bad)
json.dump(res, open(out + "/summary.json", "w"), indent=1)
PY
