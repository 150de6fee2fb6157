mkdir -p gpurun_out/r05q
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline"
for i in 1 2 3; do $B > gpurun_out/r05q/base_$i.log 2>&1; HIM_ZERO_GRAD_SIDE=0 $B > gpurun_out/r05q/nozero_$i.log 2>&1; done
for f in gpurun_out/r05q/*_[123].log; do echo $f $(tail -1 $f | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"); done
for i in 1 2; do $B --fake-comm > gpurun_out/r05q/fake_$i.log 2>&1; HIM_ADAM_CHUNKED=1 $B --fake-comm > gpurun_out/r05q/fakechunk_$i.log 2>&1; done
for f in gpurun_out/r05q/fake*.log; do echo $f $(tail -1 $f | python -c "import sys,json; d=json.loads(sys.stdin.read())['fake_comm']; print(d['ms_per_step_without'], d['ms_per_step_with'], d['delta_ms'], d['exposed_comm_ms'])" 2>&1 | tail -1); done
