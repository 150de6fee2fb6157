mkdir -p gpurun_out/r03m
python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/r03m/tests.log 2>&1; echo tests rc=$? >> gpurun_out/r03m/tests.log
bash tools/collect_profiles.sh r03m > gpurun_out/r03m/collect.log 2>&1
python bench.py > gpurun_out/r03m/bench_default.json 2> gpurun_out/r03m/bench_default.err
for w in c4 c2local box2mask; do python bench.py --workload $w --no-cpu-baseline > gpurun_out/r03m/bench_$w.json 2>/dev/null; done
tail -3 gpurun_out/r03m/tests.log; cut -c1-300 gpurun_out/r03m/bench_default.json
