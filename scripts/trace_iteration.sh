#!/bin/bash
# rocprofv3 kernel trace of three training iterations of one or more network shapes of tests/tools/bench_train_shapes.py (run on the
# GPU box through gpurun):  scripts/trace_iteration.sh OUTDIR "4x64 (config 1: 32 coarse, no fine)" "8x64" ...
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for s in "$@"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --stats -d $out/trace$i -o t --output-format csv -- python $GRAFT_REPO_ROOT/tests/tools/bench_train_shapes.py --trace-one "$s" > $out/trace$i.log 2>&1
    echo "$s" > $out/trace$i/shape.txt
    python - "$out/trace$i" <<'PY' > $out/trace$i.txt
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print(open(sys.argv[1] + "/shape.txt").read().strip(), "-- kernel time per iteration (3 iterations traced):", round(sum(float(r["TotalDurationNs"]) for r in rows) / 3e3, 1), "us")
for r in rows[:28]:
    print(f'{float(r["TotalDurationNs"]) / 3e3:9.1f} us/it {int(r["Calls"]) / 3:6.1f} calls/it  avg {float(r["AverageNs"]) / 1e3:8.1f} us  {r["Name"][:100]}')
PY
done
