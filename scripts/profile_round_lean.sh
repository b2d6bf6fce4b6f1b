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
cp $R/bench_full.json $R/gpurun_out/bench_full_$tag.json 2>/dev/null      # every object as measured (the line is the compact form)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$tag -o bench --output-format csv -- python $R/bench.py --no-cpu-baseline > $R/gpurun_out/bench_${tag}_profiled.log 2>&1
for group in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  name=$(echo $group | cut -d' ' -f1)
  rocprofv3 --pmc $group --kernel-trace -d $R/gpurun_out/pmc_$tag/$name -o pmc --output-format csv -- python $R/bench.py --headline-only --steps 1 --warmup 0 > $R/gpurun_out/pmc_$tag/$name.log 2>&1
done
cd $R
# round 6: the narrow shapes' training iterations with their stage tables, and the drop-in report (call traces of the unmodified scripts)
python tests/tools/bench_train_shapes.py --iters 10 --only "4x64 (config 1: 32 coarse, no fine),8x64,8x128,8x256" > gpurun_out/train_shapes_$tag.txt 2>/dev/null
python -m pytest tests/test_gpu_script_traces.py -q > gpurun_out/script_traces_$tag.txt 2>&1
# round 6, second half: the 64-wide networks' fused backward -- the kernel with parts compiled out (probe), its counters on a real tape,
# and config 1's iteration A/B against the separate kernels
tests/tools/probes/fb_probe 4 > gpurun_out/fb_probe_${tag}_L4.txt 2>&1
tests/tools/probes/fb_probe 8 524288 > gpurun_out/fb_probe_${tag}_L8.txt 2>&1
NM_FUSED_BACKWARD=0 python tests/tools/bench_tiny_train.py --shapes > gpurun_out/tiny_train_separate_$tag.json 2>/dev/null
python tests/tools/bench_tiny_train.py --shapes > gpurun_out/tiny_train_fused_$tag.json 2>/dev/null
for group in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "FETCH_SIZE" "WRITE_SIZE"; do
  name=$(echo $group | cut -d' ' -f1)
  bash scripts/pmc_run.sh fb_${tag}_$name "$group" "mlp_backward_dw64" python $R/tests/tools/fused_bwd_one.py > gpurun_out/pmc_fb_${tag}_$name.txt 2>&1
done
python tests/tools/fused_bwd_one.py > gpurun_out/fused_bwd_one_$tag.txt 2>&1
bash scripts/trace_iteration.sh trace_$tag "4x64 (config 1: 32 coarse, no fine)" "8x64" > /dev/null 2>&1
ls $R/gpurun_out/prof_$tag $R/gpurun_out/pmc_$tag
