#!/bin/bash
# Runs ON the GPU box: the final round-4 numbers -- profiles (tools/collect_profiles.sh) + the bench lines of every workload.
cd $GRAFT_REPO_ROOT
bash tools/collect_profiles.sh r04f > gpurun_out/r04f_collect.log 2>&1
O=gpurun_out/r04f
python bench.py --steps 20 --warmup 5 > $O/r04_bench_line.json 2> $O/bench.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --fake-comm > $O/r04_bench_line_fake_comm.json 2>> $O/bench.err
for wl in c2local c4 box2mask; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --workload $wl > $O/r04_bench_line_$wl.json 2>> $O/bench.err
done
python tools/nomfma_gaps.py $O/bench_trace.db 40 15 2 > $O/r04_nomfma_gaps.txt 2>&1
rm -f $O/bench_trace.db
for f in $O/r04_bench_line*.json; do cut -c1-140 $f; done
