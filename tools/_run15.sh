mkdir -p gpurun_out/r03n
python -m pytest tests/test_ops_gpu.py -m gpu -q -x -p no:cacheprovider -k "cond_image" 2>&1 | tail -5
b() { echo "== $*"; env "$@" python bench.py --no-cpu-baseline --no-roofline --steps 30 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
{
for rep in 1 2 3; do
b A=1
b HIM_D_SPLIT_INPUT=0
done
} > gpurun_out/r03n/ab.log 2>&1
grep -v amdgpu.ids gpurun_out/r03n/ab.log
python -m pytest tests/test_model_gpu.py -m gpu -q -x -p no:cacheprovider -k "tiny or c2_teacher or sharded or flag or sn_D" 2>&1 | tail -5
