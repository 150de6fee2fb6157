mkdir -p gpurun_out/r05n
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for T in 0 1 2; do for L in g_down1 g_down2 g_down3 g_down4 g_up2 d0_l1 d0_l3; do
  HIM_WGRAD_TILE=$T rocprofv3 --kernel-trace -d $R/gpurun_out/r05n/t_$L_$T -o r -- python $R/tools/layer_bench.py $L 5 > /dev/null 2>&1
  echo "tile=$T $L $(python $R/tools/prof_summary.py $R/gpurun_out/r05n/t_$L_$T --by-grid 2>/dev/null | grep wgrad_fast | head -1 | awk '{print $1,$2,$3,$4,$5,$6,$7,$8,$9}')"
  rm -rf $R/gpurun_out/r05n/t_$L_$T
done; done
cd $R
B="python bench.py --steps 20 --warmup 5 --no-roofline --no-cpu-baseline"
for i in 1 2; do for T in 0 1 2; do HIM_WGRAD_TILE=$T $B > gpurun_out/r05n/bench_t${T}_$i.log 2>&1; echo "bench tile=$T $(tail -1 gpurun_out/r05n/bench_t${T}_$i.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")"; done; done
