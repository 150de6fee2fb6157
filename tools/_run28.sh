mkdir -p gpurun_out/r03v
python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/r03v/tests.log 2>&1; echo tests rc=$? >> gpurun_out/r03v/tests.log
bash tools/collect_profiles.sh r03v > gpurun_out/r03v/collect.log 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/r03v/bench_default.json 2> gpurun_out/r03v/bench_default.err
for w in c4 c2local box2mask; do python bench.py --workload $w --no-cpu-baseline > gpurun_out/r03v/bench_$w.json 2>/dev/null; done
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r03v/smoke.log 2>&1
tail -3 gpurun_out/r03v/tests.log; tail -1 gpurun_out/r03v/smoke.log; cut -c1-300 gpurun_out/r03v/bench_default.json
