cd $GRAFT_REPO_ROOT
L=gpurun_out/r04e_split.log
: > $L
tools/micro/bgemm_split_micro 1024 1024 1024 20 >> $L 2>&1
tools/micro/bgemm_split_micro 1024 4096 1024 10 >> $L 2>&1
tools/micro/bgemm_split_micro 512 512 1024 20 >> $L 2>&1
cat $L
for v in 0 1; do
HIM_RESBLOCK_FUSED=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/r04e_bench_rb$v.log 2>&1; tail -1 gpurun_out/r04e_bench_rb$v.log | cut -c1-140
done
for v in 0 1; do
HIM_RESBLOCK_FUSED=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/r04e_bench_rb${v}b.log 2>&1; tail -1 gpurun_out/r04e_bench_rb${v}b.log | cut -c1-140
done
