cd $GRAFT_REPO_ROOT
for nb in 4 8; do tools/micro/onehot_micro_3 16 $nb; tools/micro/onehot_micro_0 16 $nb; done
tools/micro/onehot_micro_0 4 4; tools/micro/onehot_micro_0 64 4
python -m pytest tests/test_ops_gpu.py -m gpu -q -k "onehot" 2>&1 | tail -2
python tools/worst_kernels_bench.py 10 2>&1 | tail -1
for cfg in "" "HIM_PANEL_PIPELINE=0" "" "HIM_PANEL_PIPELINE=0"; do
  echo "== bench $cfg"
  env $cfg python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c1-130
done
python -m pytest tests/test_model_gpu.py -m gpu -q -k "schedule or tiny_traj or backward_G" 2>&1 | tail -3
