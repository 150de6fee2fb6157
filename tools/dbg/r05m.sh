mkdir -p gpurun_out/r05m
python -m pytest tests -q -m gpu --durations=15 > gpurun_out/r05m/gpu_suite.log 2>&1; grep -n "passed\|failed\|^FAILED" gpurun_out/r05m/gpu_suite.log | head; grep -n "slowest" -A12 gpurun_out/r05m/gpu_suite.log | cut -c1-120
python bench.py --steps 20 --warmup 5 > gpurun_out/r05m/bench.log 2>&1; tail -1 gpurun_out/r05m/bench.log > gpurun_out/r05m/r05_bench_line.json; cut -c1-220 gpurun_out/r05m/r05_bench_line.json
python bench.py --steps 20 --warmup 5 --fake-comm --no-cpu-baseline > gpurun_out/r05m/fake.log 2>&1; tail -1 gpurun_out/r05m/fake.log > gpurun_out/r05m/r05_bench_line_fake_comm.json
HIM_ADAM_CHUNKED=1 python bench.py --steps 20 --warmup 5 --fake-comm --no-cpu-baseline --no-roofline > gpurun_out/r05m/fake_chunked.log 2>&1
for f in fake fake_chunked; do tail -1 gpurun_out/r05m/$f.log | python -c "import sys,json; d=json.loads(sys.stdin.read())['fake_comm']; print(d['ms_per_step_without'], d['ms_per_step_with'], d['delta_ms'], d['buckets'], d['exposed_comm_ms'])"; done
python __graft_entry__.py smoke 2>&1 | tail -2
