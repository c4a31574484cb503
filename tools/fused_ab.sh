# scratch A/B script (edit per experiment)
R=$GRAFT_REPO_ROOT
for wl in cfg3 cfg2 cfg4; do
  for v in "LG_REGION_MODE=0" "LG_REGION_MODE=1" "LG_REGION_MODE=2" "LIDARGS_WORK_LISTS=0" "LG_REGION_MODE=0" "LG_REGION_MODE=1"; do
    echo "== $wl $v"; env $v python $R/tools/time_cfg.py $wl 2>&1 | tail -1 | grep -o "ms/frame [0-9.]*\|'render_bwd': [0-9.]*"  | tr '\n' ' '; echo
  done
done
