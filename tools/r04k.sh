cd $GRAFT_REPO_ROOT
for sk in 0 1 2 3 4 7; do tools/micro/onehot_micro_$sk; done
tools/micro/onehot_micro_0 4; tools/micro/onehot_micro_0 64
python -m pytest tests/test_ops_gpu.py -m gpu -q -k "onehot" 2>&1 | tail -3
python tools/worst_kernels_bench.py 10 2>&1 | tail -1
for cfg in "" "HIM_NO_ONEHOT_RLE=1" ""; do
  echo "== bench $cfg"
  env $cfg python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c1-130
done
