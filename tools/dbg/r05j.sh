mkdir -p gpurun_out/r05
bash tools/collect_profiles.sh r05 > gpurun_out/r05/collect.log 2>&1
python tools/nomfma_gaps.py gpurun_out/r05/bench_trace.db 40 > gpurun_out/r05/r05_nomfma_gaps.txt 2>&1
head -2 gpurun_out/r05/r05_nomfma_gaps.txt
bash tools/roofline_table.sh r05 > gpurun_out/r05/roofline.log 2>&1
tail -2 gpurun_out/r05/r05_roofline_table.txt
bash tools/pmc_layer.sh r05 d0_l3 gconv_fast_kernel 1128 gconv_fast_1128x4
bash tools/pmc_layer.sh r05 d0_l3 gconv_fast_kernel 540 gconv_fast_540x8
bash tools/pmc_layer.sh r05 d0_l3 wgrad_fast_kernel 32 wgrad_fast_32x4x6
ls gpurun_out/r05/*.json
python bench.py --steps 20 --warmup 5 > gpurun_out/r05/bench_line.log 2>&1; tail -1 gpurun_out/r05/bench_line.log > gpurun_out/r05/r05_bench_line.json; cut -c1-200 gpurun_out/r05/r05_bench_line.json
python bench.py --steps 20 --warmup 5 --fake-comm --no-cpu-baseline > gpurun_out/r05/bench_fake.log 2>&1; tail -1 gpurun_out/r05/bench_fake.log > gpurun_out/r05/r05_bench_line_fake_comm.json
for w in c4 c2local box2mask; do python bench.py --workload $w --steps 20 --warmup 5 > gpurun_out/r05/bench_$w.log 2>&1; tail -1 gpurun_out/r05/bench_$w.log > gpurun_out/r05/r05_bench_line_$w.json; cut -c1-160 gpurun_out/r05/r05_bench_line_$w.json; done
