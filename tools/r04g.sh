cd $GRAFT_REPO_ROOT
python -m pytest tests/test_ops_gpu.py -m gpu -q -x --durations=8 > gpurun_out/r04g_ops.log 2>&1; tail -15 gpurun_out/r04g_ops.log
for cfg in "" "HIM_NO_WINO4=1" "HIM_KEEP_WINO_INPUT=0" "" "HIM_NO_WINO4=1" "HIM_KEEP_WINO_INPUT=0"; do
  echo "== bench $cfg"
  env $cfg python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c1-130
done
python -m pytest tests/test_model_gpu.py -m gpu -q --durations=10 -k "c2_teacher or c1_teacher_forced_20 or schedule or c4_full or free_running or smoke or tiny_global_teacher" > gpurun_out/r04g_model.log 2>&1; tail -25 gpurun_out/r04g_model.log
