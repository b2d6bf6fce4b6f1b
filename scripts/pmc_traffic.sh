#!/bin/bash
# HBM traffic of the headline kernel: FETCH_SIZE and WRITE_SIZE in separate passes (kernel trace only) over one view
#   bash scripts/pmc_traffic.sh <tag>      ->  gpurun_out/pmc_<tag>/{FETCH_SIZE,WRITE_SIZE}/pmc_counter_collection.csv
set -u
tag=${1:-r02}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc_$tag
cd /tmp && export TMPDIR=/tmp
for group in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $group --kernel-trace -d $R/gpurun_out/pmc_$tag/$group -o pmc --output-format csv -- python $R/bench.py --headline-only --steps 1 --warmup 0 > $R/gpurun_out/pmc_$tag/$group.log 2>&1
done
