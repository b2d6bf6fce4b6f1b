#!/bin/bash
# Round profile on the GPU box (through gpurun):   bash scripts/profile_round.sh r03
#   1. the default bench line (unprofiled)                         -> gpurun_out/bench_<tag>.json
#   2. rocprofv3 --kernel-trace --stats of the SAME command         -> gpurun_out/prof_<tag>/bench_kernel_{stats,trace}.csv
#   3. PMC passes (one counter group per run, kernel-trace only) of `bench.py --headline-only --steps 1 --warmup 0`
#      -> gpurun_out/pmc_<tag>/<group>/pmc_counter_collection.csv
#   4. marching cubes at 480^3: per-kernel times (kernel trace) and HBM traffic (FETCH_SIZE / WRITE_SIZE passes)
#   5. the side benches (configs, shapes, mesh, weight gradients, BuFF sampler, training iteration under the kernel trace)
#   6. the 2-ranks-on-one-GPU bench line (functional multi-rank mode over gloo)
# scripts/summarize_profiles.py <tag> then writes the committed summaries under profiles/.
set -u
tag=${1:-r03}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof_$tag $R/gpurun_out/pmc_$tag
cd $R && python bench.py > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$tag -o bench --output-format csv -- python $R/bench.py > $R/gpurun_out/bench_${tag}_profiled.log 2>&1
for group in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $group | cut -d' ' -f1)
  rocprofv3 --pmc $group --kernel-trace -d $R/gpurun_out/pmc_$tag/$name -o pmc --output-format csv -- python $R/bench.py --headline-only --steps 1 --warmup 0 > $R/gpurun_out/pmc_$tag/$name.log 2>&1
done
cd $R
bash tests/tools/prof_mc.sh > gpurun_out/mc_$tag.txt 2>&1
cp gpurun_out/prof_mc/t_kernel_stats.csv gpurun_out/mc_kernel_stats_$tag.csv
for group in FETCH_SIZE WRITE_SIZE; do
  bash scripts/pmc_run.sh mc_${tag}_$group "$group" "mc_" python $R/tests/tools/bench_mesh.py --reps 2 > gpurun_out/mc_pmc_${tag}_$group.txt 2>&1
done
python scripts/bench_configs.py > gpurun_out/configs_$tag.json 2>/dev/null
python tests/tools/bench_mesh.py --reps 20 > gpurun_out/mesh_$tag.json 2>/dev/null
python tests/tools/bench_dw.py > gpurun_out/dw_$tag.json 2>/dev/null
python tests/tools/bench_shapes.py 2>/dev/null | tail -1 > gpurun_out/shapes_$tag.json
python tests/tools/bench_buff_sampler.py > gpurun_out/buff_$tag.json 2>/dev/null
(cd /tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_train_$tag -o t --output-format csv -- python $R/tests/tools/train_probe.py 20 > $R/gpurun_out/train_$tag.json 2>/dev/null)
python bench.py --gpus 1 --ranks-per-gpu 2 --steps 1 --warmup 1 > gpurun_out/bench_${tag}_2ranks.json 2> gpurun_out/bench_${tag}_2ranks.err
ls $R/gpurun_out/prof_$tag $R/gpurun_out/pmc_$tag
