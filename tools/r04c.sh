set -x
cd $GRAFT_REPO_ROOT
L=gpurun_out/r04c_micro.log
: > $L
for i in 1 2; do
tools/micro/bgemm_micro 1024 1024 1024 30 >> $L 2>&1
tools/micro/bgemm_micro_nopin 1024 1024 1024 30 >> $L 2>&1
tools/micro/gemm_micro 1024 1024 1024 30 >> $L 2>&1
done
tools/micro/bgemm_micro 1024 4096 1024 10 >> $L 2>&1
for shape in "8 64 256 512 64" "8 128 128 256 128" "8 256 64 128 256" "8 512 32 64 512" "8 512 16 32 512"; do
  echo "== $shape" >> $L
  tools/micro/wino_micro_old $shape 0 20 >> $L 2>&1
  tools/micro/wino_micro $shape 0 20 >> $L 2>&1
done
cat $L
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04c_bench.log 2>&1; tail -1 gpurun_out/r04c_bench.log | cut -c1-400
HIM_NO_BGEMM=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/r04c_bench_nobgemm.log 2>&1; tail -1 gpurun_out/r04c_bench_nobgemm.log | cut -c1-300
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --fake-comm > gpurun_out/r04c_bench_fake.log 2>&1; tail -1 gpurun_out/r04c_bench_fake.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d.get('fake_comm'))"
python tools/parity_probe.py c1_traj 4 > gpurun_out/r04c_probe.log 2>&1; tail -90 gpurun_out/r04c_probe.log
