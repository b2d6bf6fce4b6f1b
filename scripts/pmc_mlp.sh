#!/bin/bash
# PMC passes over the fused-MLP kernel variants (run on the GPU box through gpurun):
#   bash scripts/pmc_mlp.sh <tag> "<counters>" <variants...>
# one rocprofv3 --pmc run per call; the per-kernel averages are printed and kept in gpurun_out/pmc_<tag>/summary.txt
set -u
tag=$1; counters=$2; shift; shift
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $counters --kernel-trace -d $out/sq -o sq --output-format csv -- python $GRAFT_REPO_ROOT/scripts/bench_mlp.py "$@" > $out/sq.log 2>&1
grep "^variant" $out/sq.log
python - <<PY | tee -a $out/summary.txt
import csv, glob, collections
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$out/sq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "mlp_kernel" in r["Kernel_Name"] and int(r["Grid_Size"]) >= 250000:
            rows[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in rows.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    print(k)
    for c, v in sorted(m.items()):
        print(f"   {c:30s} {v:.5g}")
    wc = m.get("SQ_WAVE_CYCLES", 0)
    if wc:
        print("   fractions of SQ_WAVE_CYCLES: " + "  ".join(f"{c[3:]} {m[c] / wc:.3f}" for c in sorted(m) if c.startswith("SQ_WAIT") or c.startswith("SQ_ACTIVE")))
PY
