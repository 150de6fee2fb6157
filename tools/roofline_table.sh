#!/bin/bash
# Runs ON the GPU box: every layer of tools/layer_bench.py alone under rocprofv3 --kernel-trace (isolated per-kernel
# durations by launch grid), then tools/roofline_table.py joins them with the step trace of tools/trace_step.sh.
# Usage: bash tools/roofline_table.sh <tag>     (needs gpurun_out/<tag>/<tag>_bench_kernel_stats_by_grid.txt)
set -u
TAG=${1:-r05}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT/layers
cd /tmp && export TMPDIR=/tmp
for L in $(python $ROOT/tools/layer_bench.py --list); do
  rocprofv3 --kernel-trace -d $OUT/layers/trace_$L -o r -- python $ROOT/tools/layer_bench.py $L 5 > $OUT/layers/$L.log 2>&1
  python $ROOT/tools/prof_summary.py $OUT/layers/trace_$L --by-grid > $OUT/layers/$L.txt 2>&1
  rm -rf $OUT/layers/trace_$L
done
# in-step columns: the trace that is not launch-starved when there is one (tools/trace_unbound.sh: 12 steps), else the plain one (9)
if [ -s $OUT/${TAG}_unbound_kernel_stats_by_grid.txt ]; then
  python $ROOT/tools/roofline_table.py $OUT/layers $OUT/${TAG}_unbound_kernel_stats_by_grid.txt 12 > $OUT/${TAG}_roofline_table.txt 2>&1
else
  python $ROOT/tools/roofline_table.py $OUT/layers $OUT/${TAG}_bench_kernel_stats_by_grid.txt 9 > $OUT/${TAG}_roofline_table.txt 2>&1
fi
cd $ROOT
