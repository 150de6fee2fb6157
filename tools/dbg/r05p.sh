mkdir -p gpurun_out/r05p
F="python bench.py --steps 20 --warmup 5 --fake-comm --no-cpu-baseline --no-roofline"
for i in 1 2; do
$F > gpurun_out/r05p/fake_default_$i.log 2>&1
HIM_ZERO_GRAD_SIDE=0 $F > gpurun_out/r05p/fake_nozeroside_$i.log 2>&1
done
for f in gpurun_out/r05p/fake_*.log; do echo $f $(tail -1 $f | python -c "import sys,json; d=json.loads(sys.stdin.read())['fake_comm']; print(d['ms_per_step_without'], d['ms_per_step_with'], d['delta_ms'], d['exposed_comm_ms'])" 2>&1 | tail -1); done
python __graft_entry__.py smoke 2>&1 | tail -1
python -m pytest tests/test_model_gpu.py -q -m gpu -k "local_enhancer_full or c4_full_batch" --durations=3 2>&1 | tail -6
