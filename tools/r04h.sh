cd $GRAFT_REPO_ROOT
python -m pytest tests/test_ops_gpu.py -m gpu -q --durations=5 > gpurun_out/r04h_ops.log 2>&1; tail -8 gpurun_out/r04h_ops.log
for cfg in "" "HIM_WINO4_MIN_C=128" "HIM_VGG_BACKWARD_EARLY=1" "HIM_WINO4_MIN_C=128 HIM_VGG_BACKWARD_EARLY=1" "" "HIM_WINO4_MIN_C=128" "HIM_VGG_BACKWARD_EARLY=1" "HIM_RESBLOCK_FUSED=1"; do
  echo "== bench $cfg"
  env $cfg python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c1-130
done
