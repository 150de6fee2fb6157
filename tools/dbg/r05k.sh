mkdir -p gpurun_out/r05k
F="python bench.py --steps 20 --warmup 5 --fake-comm --no-cpu-baseline --no-roofline"
$F > gpurun_out/r05k/fake_default.log 2>&1
HIM_ZERO_GRAD_SIDE=0 $F > gpurun_out/r05k/fake_nozeroside.log 2>&1
$F --tail-mb 0 > gpurun_out/r05k/fake_notail.log 2>&1
HIM_D_FROM_IDS=0 HIM_LABEL_IDS=0 $F > gpurun_out/r05k/fake_noids.log 2>&1
HIM_ADAM_SPLIT_STEM=0 $F > gpurun_out/r05k/fake_nosplit.log 2>&1
HIM_ZERO_GRAD_SIDE=0 HIM_ADAM_SPLIT_STEM=0 HIM_D_FROM_IDS=0 HIM_LABEL_IDS=0 HIM_NO_FEWIN_FOLD=1 $F --tail-mb 0 > gpurun_out/r05k/fake_alloff.log 2>&1
for f in gpurun_out/r05k/fake_*.log; do echo $f $(tail -1 $f | python -c "import sys,json; d=json.loads(sys.stdin.read())['fake_comm']; print(d['ms_per_step_without'], d['ms_per_step_with'], d['delta_ms'], d['buckets'], d['exposed_comm_ms'])" 2>&1 | tail -1); done
bash tools/pmc_layer.sh r05 d0_l3 gconv_fast_kernel 1128 gconv_fast_1128x4
bash tools/pmc_layer.sh r05 d0_l3 gconv_fast_kernel 540 gconv_fast_540x8
bash tools/pmc_layer.sh r05 d0_l3 wgrad_fast_kernel 32 wgrad_fast_32x4x6
cat gpurun_out/r05/r05_pmc_wgrad_fast_32x4x6.json | head -30
