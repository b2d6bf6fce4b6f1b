#!/bin/bash
# per-kernel times of the marching-cubes passes at 480^3 (rocprofv3 kernel trace of tests/tools/bench_mesh.py)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_mc
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_mc -o t --output-format csv -- python $R/tests/tools/bench_mesh.py --reps 5 > $R/gpurun_out/prof_mc.json 2>/dev/null
python - <<PY
import csv, json
rows = list(csv.DictReader(open("$R/gpurun_out/prof_mc/t_kernel_stats.csv")))
for r in rows:
    if "mc_" in r["Name"] or "fill" in r["Name"]:
        print(f"{r['Name'][:58]:58s} {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:7.1f} us  min {float(r['MinNs'])/1e3:7.1f}")
d = json.load(open("$R/gpurun_out/prof_mc.json"))["marching_cubes"]
print({k: d[k] for k in ("ms_min", "ms_avg", "bitwise_identical_to_oracle", "vertices", "faces")})
PY
