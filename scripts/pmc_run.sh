#!/bin/bash
# Generic PMC pass (run on the GPU box through gpurun):
#   bash scripts/pmc_run.sh <tag> "<counters>" "<kernel name substring>" <command...>
# prints the per-kernel average of every counter; raw CSVs stay under gpurun_out/pmc_<tag>/
set -u
tag=$1; counters=$2; match=$3; shift; shift; shift
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $counters --kernel-trace -d $out/raw -o pmc --output-format csv -- "$@" > $out/cmd.log 2>&1
MATCH="$match" python - <<PY | tee -a $out/summary.txt
import csv, glob, collections, os
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$out/raw/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if os.environ["MATCH"] in r["Kernel_Name"]:
            rows[r["Kernel_Name"][:80]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in rows.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    print(k, " launches:", len(next(iter(d.values()))))
    for c, v in sorted(m.items()):
        print(f"   {c:30s} {v:.5g}")
    wc = m.get("SQ_WAVE_CYCLES", 0)
    if wc:
        print("   fractions of SQ_WAVE_CYCLES: " + "  ".join(f"{c[3:]} {m[c] / wc:.3f}" for c in sorted(m) if c.startswith("SQ_WAIT") or c.startswith("SQ_ACTIVE")))
    if wc and m.get("SQ_VALU_MFMA_BUSY_CYCLES"):
        print("   matrix pipe busy / wave-resident time (2 waves per SIMD): %.3f" % (m["SQ_VALU_MFMA_BUSY_CYCLES"] / (wc * 4 / 2)))
PY
