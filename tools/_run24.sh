mkdir -p gpurun_out/r03r
python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/r03r/tests.log 2>&1; echo tests rc=$? >> gpurun_out/r03r/tests.log
bash tools/collect_profiles.sh r03r > gpurun_out/r03r/collect.log 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/r03r/bench_default.json 2> gpurun_out/r03r/bench_default.err
for w in c4 c2local box2mask; do python bench.py --workload $w --no-cpu-baseline > gpurun_out/r03r/bench_$w.json 2>/dev/null; done
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r03r/smoke.log 2>&1
tail -3 gpurun_out/r03r/tests.log; tail -1 gpurun_out/r03r/smoke.log; cut -c1-300 gpurun_out/r03r/bench_default.json
