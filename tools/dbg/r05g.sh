mkdir -p gpurun_out/r05g
python -m pytest tests/test_ops_gpu.py -q -m gpu -k "reflect_fold or piecewise or conv2d_fwd_bwd" > gpurun_out/r05g/ops.log 2>&1; tail -3 gpurun_out/r05g/ops.log
python -m pytest tests/test_model_gpu.py -q -m gpu -k "multi_stream or label_id or backward_G_backward_D or verbatim or c2_teacher_forced or tiny_trajectories" --durations=5 > gpurun_out/r05g/model.log 2>&1; tail -12 gpurun_out/r05g/model.log
B="python bench.py --steps 20 --warmup 5 --no-roofline --no-cpu-baseline"
for i in 1 2 3; do
 $B > gpurun_out/r05g/base_$i.log 2>&1
 HIM_ADAM_SPLIT_STEM=0 $B > gpurun_out/r05g/nosplit_$i.log 2>&1
 HIM_NO_FEWIN_FOLD=1 $B > gpurun_out/r05g/nofold_$i.log 2>&1
 HIM_ZERO_GRAD_SIDE=0 $B > gpurun_out/r05g/nozeroside_$i.log 2>&1
 HIM_ADAM_SPLIT_STEM=0 HIM_NO_FEWIN_FOLD=1 HIM_ZERO_GRAD_SIDE=0 $B > gpurun_out/r05g/alloff_$i.log 2>&1
done
for f in gpurun_out/r05g/*_[123].log; do echo $f $(tail -1 $f | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" 2>&1 | tail -1); done
