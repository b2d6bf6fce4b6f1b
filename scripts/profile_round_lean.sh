#!/bin/bash
# Round profile on a 90-GPU-minute budget (through gpurun):   bash scripts/profile_round_lean.sh r04
# The evidence the bench line's numbers rest on, nothing else:
#   1. the default bench line (unprofiled)                                      -> gpurun_out/bench_<tag>.json
#   2. rocprofv3 --kernel-trace --stats of the SAME command without the CPU legs -> gpurun_out/prof_<tag>/bench_kernel_{stats,trace}.csv
#   3. PMC passes over `bench.py --headline-only --steps 1 --warmup 0`, one counter group per run (kernel trace only)
#   4. the generic-shape family: every shape's fraction (bench_shapes.py) and the matrix-pipe counters of two wide classes
# scripts/summarize_profiles.py <tag> then writes the committed summaries under profiles/.
set -u
tag=${1:-r04}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof_$tag $R/gpurun_out/pmc_$tag
cd $R && python bench.py > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$tag -o bench --output-format csv -- python $R/bench.py --no-cpu-baseline > $R/gpurun_out/bench_${tag}_profiled.log 2>&1
for group in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  name=$(echo $group | cut -d' ' -f1)
  rocprofv3 --pmc $group --kernel-trace -d $R/gpurun_out/pmc_$tag/$name -o pmc --output-format csv -- python $R/bench.py --headline-only --steps 1 --warmup 0 > $R/gpurun_out/pmc_$tag/$name.log 2>&1
done
cd $R
python tests/tools/bench_shapes.py 2>/dev/null | tail -1 > gpurun_out/shapes_$tag.json
for shape in "8 512 10" "8 320 10" "8 96 10"; do
  bash scripts/pmc_run.sh generic_${tag}_$(echo $shape | tr ' ' x) "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA" "mlp_kernel_g" python $R/tests/tools/one_shape.py $shape > gpurun_out/generic_pmc_${tag}_$(echo $shape | tr ' ' x).txt 2>&1
done
ls $R/gpurun_out/prof_$tag $R/gpurun_out/pmc_$tag
