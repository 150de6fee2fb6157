mkdir -p gpurun_out/r05e
python -m pytest tests/test_ops_gpu.py -q -m gpu -k "scalar_loss or first_patchgan" > gpurun_out/r05e/ops.log 2>&1; tail -3 gpurun_out/r05e/ops.log
python -m pytest tests/test_model_gpu.py -q -m gpu -k "direct_form_second or launcher or c4_full_batch or local_enhancer_full or c2_teacher_forced" --durations=8 > gpurun_out/r05e/model.log 2>&1; tail -14 gpurun_out/r05e/model.log
B="python bench.py --steps 20 --warmup 5 --no-roofline --no-cpu-baseline"
for i in 1 2; do
 $B > gpurun_out/r05e/base_$i.log 2>&1
 HIM_REAL_VGG_FIRST=1 $B > gpurun_out/r05e/vggfirst_$i.log 2>&1
 HIM_WINO_MIN_C=128 HIM_WINO_FUSED_MAX_C=127 $B > gpurun_out/r05e/wino128_$i.log 2>&1
 HIM_WINO_MIN_C=128 HIM_WINO_FUSED_MAX_C=127 HIM_WINO4_MIN_C=128 $B > gpurun_out/r05e/wino128_w4_$i.log 2>&1
 HIM_ZERO_GRAD_SIDE=1 $B > gpurun_out/r05e/zeroside_$i.log 2>&1
 HIM_REAL_VGG_FIRST=1 HIM_ZERO_GRAD_SIDE=1 $B > gpurun_out/r05e/vggfirst_zeroside_$i.log 2>&1
done
for f in gpurun_out/r05e/*_[12].log; do echo $f $(tail -1 $f | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" 2>&1 | tail -1); done
