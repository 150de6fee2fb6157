mkdir -p gpurun_out/r05r
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline"
$B --fake-comm > gpurun_out/r05r/fake.log 2>&1; tail -1 gpurun_out/r05r/fake.log > gpurun_out/r05r/r05_bench_line_fake_comm.json
HIM_ADAM_CHUNKED=1 $B --no-roofline --fake-comm > gpurun_out/r05r/fake_chunked.log 2>&1
$B --no-roofline --fake-comm --g-backward-first > gpurun_out/r05r/fake_gfirst.log 2>&1
$B --no-roofline --fake-comm --tail-mb 0 > gpurun_out/r05r/fake_notail.log 2>&1
for f in fake fake_chunked fake_gfirst fake_notail; do echo $f $(tail -1 gpurun_out/r05r/$f.log | python -c "import sys,json; d=json.loads(sys.stdin.read())['fake_comm']; print(d['ms_per_step_without'], d['ms_per_step_with'], d['delta_ms'], d['buckets'], d['exposed_comm_ms'])" 2>&1 | tail -1); done
