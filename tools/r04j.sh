cd $GRAFT_REPO_ROOT
python -m pytest tests/test_ops_gpu.py -m gpu -q -k "onehot" > gpurun_out/r04j_ops.log 2>&1; tail -6 gpurun_out/r04j_ops.log
python tools/worst_kernels_bench.py 10 2>&1 | tail -2
HIM_NO_ONEHOT_RLE=1 python tools/worst_kernels_bench.py 10 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/wk -o r -- python $GRAFT_REPO_ROOT/tools/worst_kernels_bench.py 5 > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/prof_summary.py /tmp/wk 2>/dev/null | head -12 | cut -c1-150
cd $GRAFT_REPO_ROOT
for cfg in "" "HIM_NO_ONEHOT_RLE=1" ""; do
  echo "== bench $cfg"
  env $cfg python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c1-130
done
