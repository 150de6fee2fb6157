#!/bin/bash
# Runs ON the GPU box: rocprofv3 kernel trace of the bench command only (part 1 of tools/collect_profiles.sh).
# Usage: bash tools/trace_step.sh <tag> [extra bench.py flags]     -> gpurun_out/<tag>/<tag>_bench_*.txt
set -u
TAG=${1:-r04}
shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/bench_trace -o r -- python $ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline "$@" > $OUT/bench_under_trace.log 2>&1
python $ROOT/tools/prof_summary.py $OUT/bench_trace > $OUT/${TAG}_bench_kernel_stats.txt 2>&1
python $ROOT/tools/prof_summary.py $OUT/bench_trace --by-grid > $OUT/${TAG}_bench_kernel_stats_by_grid.txt 2>&1
python $ROOT/tools/stream_view.py $OUT/bench_trace > $OUT/${TAG}_bench_stream_view.txt 2>&1
python $ROOT/tools/timeline.py $OUT/bench_trace > $OUT/${TAG}_bench_timeline.txt 2>&1
for f in $(find $OUT/bench_trace -name "*.db" -size -30M); do cp $f $OUT/bench_trace.db; done
python $ROOT/tools/step_phases.py $OUT/bench_trace.db 80 > $OUT/${TAG}_bench_step_phases.txt 2>&1
rm -rf $OUT/bench_trace
cd $ROOT
