#!/bin/bash
# Runs ON the GPU box (gpurun): kernel traces + PMC passes behind the numbers quoted in DESIGN.md / bench.py.
# Usage: bash tools/collect_profiles.sh <tag>      -> gpurun_out/<tag>/...   (copy the summaries into profiles/)
set -u
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# 0. the schedule from a trace that is not launch-starved (timeline, stream view, step phases, gap map, by-grid stats); its
#    database also gives the dominant launch's IN-STEP durations (step 2)
bash $ROOT/tools/trace_unbound.sh $TAG
cd /tmp
# 1. the bench command under the kernel trace
rocprofv3 --kernel-trace -d $OUT/bench_trace -o r -- python $ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline > $OUT/bench_under_trace.log 2>&1
python $ROOT/tools/prof_summary.py $OUT/bench_trace > $OUT/${TAG}_bench_kernel_stats.txt 2>&1
python $ROOT/tools/prof_summary.py $OUT/bench_trace --by-grid > $OUT/${TAG}_bench_kernel_stats_by_grid.txt 2>&1
python $ROOT/tools/stream_view.py $OUT/bench_trace > $OUT/${TAG}_bench_stream_view.txt 2>&1
python $ROOT/tools/timeline.py $OUT/bench_trace > $OUT/${TAG}_bench_timeline.txt 2>&1
for f in $(find $OUT/bench_trace -name "*.db" -size -30M); do cp $f $OUT/bench_trace.db; done
python $ROOT/tools/step_phases.py $OUT/bench_trace.db 80 > $OUT/${TAG}_bench_step_phases.txt 2>&1
rm -rf $OUT/bench_trace
# 2. the dominant launch on its own: kernel trace with stats, then one PMC group per pass
rocprofv3 --kernel-trace --stats -d $OUT/gemm_trace -o r -- python $ROOT/tools/gemm_bench.py 20 > $OUT/gemm_under_trace.log 2>&1
python $ROOT/tools/prof_summary.py $OUT/gemm_trace > $OUT/${TAG}_gemm_bench_kernel_stats.txt 2>&1
python $ROOT/tools/dominant_kernel_json.py $OUT/gemm_trace $OUT/ub_trace.db $OUT/${TAG}_dominant_kernel_rocprof.json ${TAG}_unbound > $OUT/dominant_kernel_json.log 2>&1
rm -rf $OUT/gemm_trace
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_INST_LDS"; do
  d=$OUT/pmc_$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --kernel-trace -f csv -d $d -- python $ROOT/tools/gemm_bench.py 10 > $d.log 2>&1
done
python $ROOT/tools/pmc_collect.py bgemm_kernel 1024 $OUT/${TAG}_pmc_dominant_kernel.json $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_TCC_HIT_sum $OUT/pmc_SQ_WAVE_CYCLES > $OUT/pmc_collect.log 2>&1
rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_TCC_HIT_sum $OUT/pmc_SQ_WAVE_CYCLES
# 3. the two few-channel weight gradients VERDICT r2 named (head 64->3, one-hot stem): kernel trace + FETCH / WRITE passes
python $ROOT/tools/worst_kernels_bench.py 10 > $OUT/worst_kernels_bench.log 2>&1
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
  d=$OUT/pmcw_$grp
  rocprofv3 --pmc $grp --kernel-trace -f csv -d $d -- python $ROOT/tools/worst_kernels_bench.py 5 > $d.log 2>&1
done
python $ROOT/tools/pmc_collect.py wgrad_fewch_mfma_kernel 0 $OUT/${TAG}_pmc_wgrad_fewch.json $OUT/pmcw_FETCH_SIZE $OUT/pmcw_WRITE_SIZE > $OUT/pmcw_collect.log 2>&1
python $ROOT/tools/pmc_collect.py onehot_wgrad_rle_kernel 0 $OUT/${TAG}_pmc_onehot_wgrad.json $OUT/pmcw_FETCH_SIZE $OUT/pmcw_WRITE_SIZE >> $OUT/pmcw_collect.log 2>&1
rm -rf $OUT/pmcw_FETCH_SIZE $OUT/pmcw_WRITE_SIZE
# 4. the fused Winograd kernel (VGG conv1_2 shape: 8 x 64 x 256 x 512 -> 64; round 6: the persistent kernel, grid 256), stand-alone microbenchmark binary
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  d=$OUT/pmcf_$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --kernel-trace -f csv -d $d -- $ROOT/tools/micro/wino_micro 8 64 256 512 64 0 5 2 > $d.log 2>&1   # 2 = the persistent kernel (round 6)
done
python $ROOT/tools/pmc_collect.py wino_fused2_kernel 0 $OUT/${TAG}_pmc_wino_fused.json $OUT/pmcf_FETCH_SIZE $OUT/pmcf_WRITE_SIZE $OUT/pmcf_TCC_HIT_sum $OUT/pmcf_SQ_WAVE_CYCLES $OUT/pmcf_SQ_LDS_BANK_CONFLICT > $OUT/pmcf_collect.log 2>&1
rm -rf $OUT/pmcf_FETCH_SIZE $OUT/pmcf_WRITE_SIZE $OUT/pmcf_TCC_HIT_sum $OUT/pmcf_SQ_WAVE_CYCLES $OUT/pmcf_SQ_LDS_BANK_CONFLICT
cd $ROOT
