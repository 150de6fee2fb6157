cd $GRAFT_REPO_ROOT
for wl in box2mask c4; do
for cfg in "" "HIM_WINO_MIN_C=256 HIM_WINO_FUSED_MAX_C=255" "HIM_WINO_MIN_C=128 HIM_WINO_FUSED_MAX_C=127"; do
  echo "== $wl $cfg"
  env $cfg python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --workload $wl 2>&1 | tail -1 | cut -c1-150
done; done
