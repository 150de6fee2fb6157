mkdir -p gpurun_out/r05s
python -m pytest tests/test_ops_gpu.py -q -m gpu -k "opt_in_f4x4 or winograd_f4x4 or cached_panel" 2>&1 | tail -4
B="python bench.py --steps 20 --warmup 5"
for i in 1 2; do $B --no-roofline --no-cpu-baseline > gpurun_out/r05s/base_$i.log 2>&1; $B --no-roofline --no-cpu-baseline --variant f4x4-resblock-fwd > gpurun_out/r05s/f4_$i.log 2>&1; done
for f in gpurun_out/r05s/*_[12].log; do echo $f $(tail -1 $f | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('variant'))"); done
$B --variant f4x4-resblock-fwd > gpurun_out/r05s/f4_full.log 2>&1; tail -1 gpurun_out/r05s/f4_full.log > gpurun_out/r05s/r05_bench_line_f4x4_resblock_fwd.json; cut -c1-300 gpurun_out/r05s/r05_bench_line_f4x4_resblock_fwd.json
python tools/variant_parity.py f4x4-resblock-fwd > gpurun_out/r05s/variant_parity.log 2>&1; tail -60 gpurun_out/r05s/variant_parity.log | cut -c1-220
