#!/bin/bash
# Runs ON the GPU box: the training step's schedule from a trace that is NOT launch-starved.  rocprofv3's per-dispatch
# interception makes the host 2.8x slower (tools/host_probe.py: 56 instead of 20 ms per step) -- slower than the device,
# so a plain traced run shows the profiler's gaps.  Here the GPU is held behind a spin kernel while the host enqueues 8
# steps (tools/host_probe.py <steps> <hold_ms>), then runs them with every launch already queued.
# Usage: bash tools/trace_unbound.sh <tag>     -> gpurun_out/<tag>/<tag>_unbound_*.txt
set -u
TAG=${1:-r05}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/ub_trace -o r -- python $ROOT/tools/host_probe.py 8 600 > $OUT/ub_probe.log 2>&1
grep host_ms $OUT/ub_probe.log > $OUT/${TAG}_unbound_host_probe.json
for f in $(find $OUT/ub_trace -name "*.db" -size -30M); do cp $f $OUT/ub_trace.db; done
python $ROOT/tools/timeline.py $OUT/ub_trace > $OUT/${TAG}_unbound_timeline.txt 2>&1
python $ROOT/tools/stream_view.py $OUT/ub_trace > $OUT/${TAG}_unbound_stream_view.txt 2>&1
python $ROOT/tools/step_phases.py $OUT/ub_trace.db 80 > $OUT/${TAG}_unbound_step_phases.txt 2>&1
python $ROOT/tools/nomfma_gaps.py $OUT/ub_trace.db > $OUT/${TAG}_unbound_nomfma_gaps.txt 2>&1
python $ROOT/tools/prof_summary.py $OUT/ub_trace --by-grid > $OUT/${TAG}_unbound_kernel_stats_by_grid.txt 2>&1
rm -rf $OUT/ub_trace
cd $ROOT
