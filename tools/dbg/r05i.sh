mkdir -p gpurun_out/r05i
B="python bench.py --steps 20 --warmup 5 --no-roofline --no-cpu-baseline"
for i in 1 2 3; do
 $B > gpurun_out/r05i/base_$i.log 2>&1
 HIM_WINO_FUSED_CHUNK=4 $B > gpurun_out/r05i/ck4_$i.log 2>&1
done
for f in gpurun_out/r05i/*_[123].log; do echo $f $(tail -1 $f | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" 2>&1 | tail -1); done
for w in c4 c2local box2mask; do
 python bench.py --workload $w --steps 20 --warmup 5 --no-roofline --no-cpu-baseline > gpurun_out/r05i/${w}_base.log 2>&1
 HIM_WINO_FUSED_CHUNK=4 python bench.py --workload $w --steps 20 --warmup 5 --no-roofline --no-cpu-baseline > gpurun_out/r05i/${w}_ck4.log 2>&1
 for v in base ck4; do echo $w $v $(tail -1 gpurun_out/r05i/${w}_$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" 2>&1 | tail -1); done
done
python -m pytest tests/test_ops_gpu.py -q -m gpu -k "winograd or vgg_loss or gated" > gpurun_out/r05i/ops.log 2>&1; tail -3 gpurun_out/r05i/ops.log
