mkdir -p gpurun_out/r05l
python -m pytest tests/test_model_gpu.py -q -m gpu -k "multi_stream or backward_G_backward_D or verbatim or two_ranks or sharded or rccl_reducer or launcher" > gpurun_out/r05l/model.log 2>&1; tail -4 gpurun_out/r05l/model.log
B="python bench.py --steps 20 --warmup 5 --no-roofline --no-cpu-baseline"
for i in 1 2 3; do $B > gpurun_out/r05l/base_$i.log 2>&1; HIM_ZERO_GRAD_SIDE=0 $B > gpurun_out/r05l/nozero_$i.log 2>&1; done
for f in gpurun_out/r05l/*_[123].log; do echo $f $(tail -1 $f | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" 2>&1 | tail -1); done
$B --fake-comm > gpurun_out/r05l/fake.log 2>&1; tail -1 gpurun_out/r05l/fake.log | python -c "import sys,json; d=json.loads(sys.stdin.read())['fake_comm']; print(d['ms_per_step_without'], d['ms_per_step_with'], d['delta_ms'], d['buckets'], d['exposed_comm_ms'])"
HIM_ADAM_CHUNKED=1 $B --fake-comm > gpurun_out/r05l/fake_chunked.log 2>&1; tail -1 gpurun_out/r05l/fake_chunked.log | python -c "import sys,json; d=json.loads(sys.stdin.read())['fake_comm']; print(d['ms_per_step_without'], d['ms_per_step_with'], d['delta_ms'])"
